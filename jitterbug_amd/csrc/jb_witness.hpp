// jb_witness.hpp — run-time witness for the geom pairs the simulator does NOT collide.
//
// Every jitterbug geom has contype = conaffinity = 1 (reference jitterbug.xml:44-107), so MuJoCo tests the 160 geom pairs whose bodies
// differ and are not parent and child.  The step kernels collide every geom with the floor and, of those 160, the eight pairs that do
// occur on randomised models (mass ellipsoid / motor-axis thread against the upper-leg cylinders).  For the reference's model
// and the reference's randomisation the other 152 never touch (tests/test_gpu_clearance.py) - but sigmas are one keyword away.  This pass
// tells a user whose models leave the tested distribution: per environment, the smallest distance over the 152 unsimulated pairs in the
// current state, exact (GJK on the four primitive types, fp64), from the very constant tables the step kernel reads.  One thread per
// environment, a diagnostic kernel off the hot path (JB_FLAG_PAIR_WITNESS / jb_pair_witness).
//
// Written once for the device and for the host test harness (tests/host_harness.cpp holds it against oracle/jb_clearance.c).
#pragma once
#include <cmath>

#include "jb_sim.hpp"

namespace jb {

struct WGeom { int type, body; double c[3], R[9], s[3], rb; };      // type: 0 sphere, 1 cylinder (axis = column 2 of R), 2 box, 3 ellipsoid; pose in ROOT coordinates
constexpr int W_NGEOM = 22;

JB_HD void w_mv(double* o, const double* R, const double* v) {
    const double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2], z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
JB_HD void w_mtv(double* o, const double* R, const double* v) {
    const double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2], z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
JB_HD void w_mm(double* o, const double* A, const double* B) {
    double t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    for (int i = 0; i < 9; i++) o[i] = t[i];
}
JB_HD double w_dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
JB_HD void w_rodrigues(double* R, const double* e, double th) {
    const double c = cos(th), s = sin(th), v = 1 - c;
    R[0] = c + e[0] * e[0] * v;        R[1] = e[0] * e[1] * v - e[2] * s; R[2] = e[0] * e[2] * v + e[1] * s;
    R[3] = e[1] * e[0] * v + e[2] * s; R[4] = c + e[1] * e[1] * v;        R[5] = e[1] * e[2] * v - e[0] * s;
    R[6] = e[2] * e[0] * v - e[1] * s; R[7] = e[2] * e[1] * v + e[0] * s; R[8] = c + e[2] * e[2] * v;
}
// a rotation whose third column is the unit vector `ax` (a cylinder needs nothing else of its frame)
JB_HD void w_frame_from_axis(double* R, const double* ax) {
    const int k = fabs(ax[0]) < fabs(ax[1]) ? (fabs(ax[0]) < fabs(ax[2]) ? 0 : 2) : (fabs(ax[1]) < fabs(ax[2]) ? 1 : 2);
    double u[3] = {0, 0, 0}, v[3];
    u[k] = 1.0;
    const double d = u[0] * ax[0] + u[1] * ax[1] + u[2] * ax[2];
    double n = 0;
    for (int i = 0; i < 3; i++) { u[i] -= d * ax[i]; n += u[i] * u[i]; }
    n = 1.0 / sqrt(n);
    for (int i = 0; i < 3; i++) u[i] *= n;
    v[0] = ax[1] * u[2] - ax[2] * u[1]; v[1] = ax[2] * u[0] - ax[0] * u[2]; v[2] = ax[0] * u[1] - ax[1] * u[0];
    for (int i = 0; i < 3; i++) { R[3 * i] = u[i]; R[3 * i + 1] = v[i]; R[3 * i + 2] = ax[i]; }
}

// The 22 geoms in root coordinates from a packed constant table (LM_TABLE layout: what the step kernel reads) and the joint angles.
// Geom numbering follows the reference XML like the parameter table's: 0, 1 core boxes, 2 screw cylinder, 3 screw ellipsoid, 4 + 4 l ...
// 7 + 4 l upper cylinder / knee tip / lower cylinder / foot of leg l, 20 motor-axis thread, 21 eccentric mass.
template <typename R_>
JB_HD void witness_geoms(const R_* tab, const double (&th1)[4], const double (&th2)[4], double phi, WGeom* g) {
    auto L = [&](int i, int leg) { return (double)tab[lm_offset(i, leg)]; };
    auto L3 = [&](int i, int leg, double* o) { for (int k = 0; k < 3; k++) o[k] = L(i + k, leg); };
    auto set_cyl = [&](WGeom& q, int body, const double* c, const double* ax, double r, double h) {
        q.type = 1; q.body = body;
        for (int k = 0; k < 3; k++) q.c[k] = c[k];
        w_frame_from_axis(q.R, ax);
        q.s[0] = r; q.s[1] = h; q.s[2] = 0; q.rb = sqrt(r * r + h * h);
    };
    auto set_sph = [&](WGeom& q, int body, const double* c, double r) {
        q.type = 0; q.body = body;
        for (int k = 0; k < 3; k++) { q.c[k] = c[k]; q.s[k] = r; }
        for (int k = 0; k < 9; k++) q.R[k] = (k % 4 == 0) ? 1.0 : 0.0;
        q.rb = r;
    };
    double am[3], em[3], Rm[9];
    L3(LM_AM, 0, am); L3(LM_EM, 0, em);
    w_rodrigues(Rm, em, phi);
    for (int leg = 0; leg < 4; leg++) {
        double a1[3], e1[3], E2[3], R1[9], R2[9], R12[9], t[3], v[3], a2[3], c[3], ax[3];
        L3(LM_A1, leg, a1); L3(LM_E1, leg, e1); L3(LM_E2, leg, E2);
        w_rodrigues(R1, e1, th1[leg]); w_rodrigues(R2, E2, th2[leg]);
        w_mm(R12, R1, R2);
        L3(LM_DA2, leg, t); w_mv(v, R1, t);
        for (int k = 0; k < 3; k++) a2[k] = a1[k] + v[k];
        const int bu = 1 + 2 * leg, bl = 2 + 2 * leg;
        L3(LM_UC_D, leg, t); w_mv(v, R1, t); for (int k = 0; k < 3; k++) c[k] = a1[k] + v[k];
        L3(LM_UC_AX, leg, t); w_mv(ax, R1, t);
        set_cyl(g[4 + 4 * leg], bu, c, ax, L(LM_UC_R, leg), L(LM_UC_H, leg));
        L3(LM_DTIP, leg, t); w_mv(v, R1, t); for (int k = 0; k < 3; k++) c[k] = a1[k] + v[k];
        set_sph(g[5 + 4 * leg], bu, c, L(LM_TIP_R, leg));
        L3(LM_LC_D, leg, t); w_mv(v, R12, t); for (int k = 0; k < 3; k++) c[k] = a2[k] + v[k];
        L3(LM_LC_AX, leg, t); w_mv(ax, R12, t);
        set_cyl(g[6 + 4 * leg], bl, c, ax, L(LM_LC_R, leg), L(LM_LC_H, leg));
        L3(LM_DFOOT, leg, t); w_mv(v, R12, t); for (int k = 0; k < 3; k++) c[k] = a2[k] + v[k];
        set_sph(g[7 + 4 * leg], bl, c, L(LM_FOOT_R, leg));
        // the lane's root / motor-body geoms
        if (leg < 2) {
            WGeom& q = g[leg];
            q.type = 2; q.body = 0;
            L3(LM_XB_C, leg, q.c); L3(LM_XB_S, leg, q.s);
            for (int k = 0; k < 9; k++) q.R[k] = L(LM_XB_R + k, leg);
            q.rb = sqrt(q.s[0] * q.s[0] + q.s[1] * q.s[1] + q.s[2] * q.s[2]);
        } else {
            const bool onm = leg == 3;
            double cc[3], cax[3], Re[9];
            L3(LM_XC_C, leg, cc); L3(LM_XC_AX, leg, cax);
            WGeom& qe = g[onm ? 21 : 3];
            qe.type = 3; qe.body = onm ? 9 : 0;
            L3(LM_XE_C, leg, qe.c); L3(LM_XE_S, leg, qe.s);
            for (int k = 0; k < 9; k++) Re[k] = L(LM_XE_R + k, leg);
            if (onm) {
                for (int k = 0; k < 3; k++) t[k] = cc[k] - am[k];
                w_mv(v, Rm, t); for (int k = 0; k < 3; k++) cc[k] = am[k] + v[k];
                w_mv(v, Rm, cax); for (int k = 0; k < 3; k++) cax[k] = v[k];
                for (int k = 0; k < 3; k++) t[k] = qe.c[k] - am[k];
                w_mv(v, Rm, t); for (int k = 0; k < 3; k++) qe.c[k] = am[k] + v[k];
                w_mm(qe.R, Rm, Re);
            } else {
                for (int k = 0; k < 9; k++) qe.R[k] = Re[k];
            }
            qe.rb = fmax(qe.s[0], fmax(qe.s[1], qe.s[2]));
            set_cyl(g[onm ? 20 : 2], onm ? 9 : 0, cc, cax, L(LM_XC_R, leg), L(LM_XC_H, leg));
        }
    }
}

// support point of a geom in direction d (spheres enter GJK as points; their radii are subtracted at the end)
JB_HD void w_support(const WGeom& g, const double* d, double* out) {
    double dl[3], sl[3] = {0, 0, 0};
    w_mtv(dl, g.R, d);
    const double* s = g.s;
    if (g.type == 1) {
        const double r = sqrt(dl[0] * dl[0] + dl[1] * dl[1]);
        if (r > 1e-300) { sl[0] = s[0] * dl[0] / r; sl[1] = s[0] * dl[1] / r; }
        sl[2] = dl[2] >= 0 ? s[1] : -s[1];
    } else if (g.type == 2) {
        for (int i = 0; i < 3; i++) sl[i] = dl[i] >= 0 ? s[i] : -s[i];
    } else if (g.type == 3) {
        const double den = sqrt(s[0] * s[0] * dl[0] * dl[0] + s[1] * s[1] * dl[1] * dl[1] + s[2] * s[2] * dl[2] * dl[2]);
        if (den > 1e-300) for (int i = 0; i < 3; i++) sl[i] = s[i] * s[i] * dl[i] / den;
    }
    w_mv(out, g.R, sl);
    for (int i = 0; i < 3; i++) out[i] += g.c[i];
}

// closest point of a simplex (1-4 vertices, rows of W) to the origin; the sub-simplex that carries it is compacted to the front.
// Returns the new vertex count (0: a proper tetrahedron contains the origin).  Exhaustive over the <= 15 sub-simplices, Gram systems.
JB_HD int w_closest_simplex(double (&W)[4][3], int n, double* v) {
    int best_mask = 0, best_n = 0;
    double best_d2 = 1e300, best_lam[4] = {0, 0, 0, 0};
    for (int mask = 1; mask < (1 << n); mask++) {
        int idx[4], m = 0;
        for (int i = 0; i < n; i++) if (mask >> i & 1) idx[m++] = i;
        double lam[4] = {1, 0, 0, 0};
        if (m > 1) {
            double E[3][3], G[3][3], r[3], mu[3] = {0, 0, 0};
            for (int k = 1; k < m; k++) for (int c = 0; c < 3; c++) E[k - 1][c] = W[idx[k]][c] - W[idx[0]][c];
            for (int a = 0; a < m - 1; a++) { r[a] = -w_dot(E[a], W[idx[0]]); for (int b = 0; b < m - 1; b++) G[a][b] = w_dot(E[a], E[b]); }
            bool ok = true;
            if (m == 2) { if (G[0][0] > 0) mu[0] = r[0] / G[0][0]; else ok = false; }
            else if (m == 3) {
                const double det = G[0][0] * G[1][1] - G[0][1] * G[1][0];
                if (fabs(det) > 1e-300) { mu[0] = (r[0] * G[1][1] - r[1] * G[0][1]) / det; mu[1] = (G[0][0] * r[1] - G[1][0] * r[0]) / det; } else ok = false;
            } else {
                const double det = G[0][0] * (G[1][1] * G[2][2] - G[1][2] * G[2][1]) - G[0][1] * (G[1][0] * G[2][2] - G[1][2] * G[2][0]) + G[0][2] * (G[1][0] * G[2][1] - G[1][1] * G[2][0]);
                if (fabs(det) > 1e-300) {
                    mu[0] = (r[0] * (G[1][1] * G[2][2] - G[1][2] * G[2][1]) - G[0][1] * (r[1] * G[2][2] - G[1][2] * r[2]) + G[0][2] * (r[1] * G[2][1] - G[1][1] * r[2])) / det;
                    mu[1] = (G[0][0] * (r[1] * G[2][2] - G[1][2] * r[2]) - r[0] * (G[1][0] * G[2][2] - G[1][2] * G[2][0]) + G[0][2] * (G[1][0] * r[2] - r[1] * G[2][0])) / det;
                    mu[2] = (G[0][0] * (G[1][1] * r[2] - r[1] * G[2][1]) - G[0][1] * (G[1][0] * r[2] - r[1] * G[2][0]) + r[0] * (G[1][0] * G[2][1] - G[1][1] * G[2][0])) / det;
                } else ok = false;
            }
            if (!ok) continue;
            lam[0] = 1;
            for (int k = 1; k < m; k++) { lam[k] = mu[k - 1]; lam[0] -= mu[k - 1]; }
            bool neg = false;
            for (int k = 0; k < m; k++) if (lam[k] < -1e-14) neg = true;
            if (neg) continue;
        }
        double p[3] = {0, 0, 0};
        for (int k = 0; k < m; k++) for (int c = 0; c < 3; c++) p[c] += lam[k] * W[idx[k]][c];
        const double d2 = w_dot(p, p);
        if (m == 4 && d2 > 1e-22) continue;      // a (nearly flat) tetrahedron only counts when it really contains the origin
        if (d2 < best_d2) { best_d2 = d2; best_mask = mask; best_n = m; for (int k = 0; k < 4; k++) best_lam[k] = k < m ? lam[k] : 0.0; }
    }
    double Wn[4][3];
    int m = 0;
    v[0] = v[1] = v[2] = 0;
    for (int i = 0; i < n; i++) if (best_mask >> i & 1) { for (int c = 0; c < 3; c++) { Wn[m][c] = W[i][c]; v[c] += best_lam[m] * W[i][c]; } m++; }
    for (int i = 0; i < m; i++) for (int c = 0; c < 3; c++) W[i][c] = Wn[i][c];
    if (best_n == 4) return 0;
    return m;
}
// distance between two convex geoms (0 when they intersect)
JB_HD double w_gjk_distance(const WGeom& a, const WGeom& b) {
    const double ra = a.type == 0 ? a.s[0] : 0.0, rb = b.type == 0 ? b.s[0] : 0.0;
    double W[4][3], v[3], sa[3], sb[3], w[3], mv[3];
    int n = 0;
    for (int i = 0; i < 3; i++) v[i] = a.c[i] - b.c[i];
    if (w_dot(v, v) < 1e-30) { v[0] = 1; v[1] = v[2] = 0; }
    for (int it = 0; it < 200; it++) {
        for (int i = 0; i < 3; i++) mv[i] = -v[i];
        if (a.type == 0) { for (int i = 0; i < 3; i++) sa[i] = a.c[i]; } else w_support(a, mv, sa);
        if (b.type == 0) { for (int i = 0; i < 3; i++) sb[i] = b.c[i]; } else w_support(b, v, sb);
        for (int i = 0; i < 3; i++) w[i] = sa[i] - sb[i];
        const double vv = w_dot(v, v), vw = w_dot(v, w);
        if (it > 0 && vv - vw <= 1e-12 * vv + 1e-24) break;          // no further progress towards the origin
        for (int i = 0; i < 3; i++) W[n][i] = w[i];
        n++;
        n = w_closest_simplex(W, n, v);
        if (n == 0 || w_dot(v, v) < 1e-24) return 0.0;               // the cores intersect
    }
    const double d = sqrt(w_dot(v, v)) - ra - rb;
    return d > 0 ? d : 0.0;
}
JB_HD bool w_pair_tested(int bi, int bj) {
    // bodies: 0 root, 1 + 2 l / 2 + 2 l upper / lower leg l, 9 motor; parents: upper -> root, lower -> its upper, motor -> root
    auto parent = [](int b) { return b == 0 ? -1 : b == 9 ? 0 : (b & 1) ? 0 : b - 1; };
    return bi != bj && parent(bi) != bj && parent(bj) != bi;
}
// the simulated geom-geom pairs: thread (20) and mass (21) against the four upper-leg cylinders
JB_HD bool w_pair_simulated(int i, int j) { return (j == 20 || j == 21) && i >= 4 && i < 20 && ((i - 4) & 3) == 0; }

// smallest distance over the UNSIMULATED pairs MuJoCo would test (0: some pair interpenetrates); pair: the geoms that attain it
template <typename R_>
JB_HD double witness_clearance(const R_* tab, const double (&th1)[4], const double (&th2)[4], double phi, int* pair) {
    WGeom g[W_NGEOM];
    witness_geoms<R_>(tab, th1, th2, phi, g);
    double best = 1e300;
    int bi = -1, bj = -1;
    for (int i = 0; i < W_NGEOM; i++)
        for (int j = i + 1; j < W_NGEOM; j++) {
            if (!w_pair_tested(g[i].body, g[j].body) || w_pair_simulated(i, j)) continue;
            const double dc[3] = {g[i].c[0] - g[j].c[0], g[i].c[1] - g[j].c[1], g[i].c[2] - g[j].c[2]};
            if (sqrt(w_dot(dc, dc)) - g[i].rb - g[j].rb >= best) continue;          // bounding spheres cannot beat the minimum
            const double d = w_gjk_distance(g[i], g[j]);
            if (d < best) { best = d; bi = i; bj = j; }
        }
    if (pair) { pair[0] = bi; pair[1] = bj; }
    return best;
}

}  // namespace jb

/* jb_clearance.c — TEST INFRASTRUCTURE (part of the oracle library): exact distances between the model's geoms.
 *
 * MuJoCo tests every geom pair whose bodies differ and are not parent and child (filterparent) when both have
 * contype/conaffinity set - in the reference model (jitterbug.xml:44-107: every jitterbug geom has the default
 * contype = conaffinity = 1; only the target opts out, :115-116) that is: root <-> lower legs, mass body <-> every leg
 * body, and every leg body <-> the bodies of the other legs.  The HIP kernel and the oracle's substep collide geoms with
 * the FLOOR only.  This file measures what that leaves out: the minimum distance over all those pairs in a given
 * configuration, by GJK on the geoms' support functions (sphere, cylinder, box, ellipsoid - the model's four types).  A
 * positive minimum over a rollout means MuJoCo would have generated no geom-geom contact there either.
 *
 * jbo_pair_clearance:        one configuration -> minimum distance and the pair that attains it
 * jbo_pair_clearance_batch:  n configurations (OpenMP), shared or per-env parameter tables
 */
#include <math.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/jitterbug_model.h"

#define NB JB_NBODY
static const int PARENT_[NB] = {-1, 0, 1, 0, 3, 0, 5, 0, 7, 0};

typedef struct { int type; double c[3], R[9], s[3], rb; } Geom;     /* world pose; rb: bounding-sphere radius */

static void mv3(double* o, const double* R, const double* v) {
    double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2], y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2], z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static void mtv3(double* o, const double* R, const double* v) {
    double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2], y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2], z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static void mm3(double* o, const double* A, const double* B) {
    double t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(o, t, sizeof t);
}
static double dot_(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static void cross_(double* o, const double* a, const double* b) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static void rodrigues_(double* R, const double* e, double th) {
    double c = cos(th), s = sin(th), v = 1 - c;
    R[0] = c + e[0] * e[0] * v;        R[1] = e[0] * e[1] * v - e[2] * s; R[2] = e[0] * e[2] * v + e[1] * s;
    R[3] = e[1] * e[0] * v + e[2] * s; R[4] = c + e[1] * e[1] * v;        R[5] = e[1] * e[2] * v - e[0] * s;
    R[6] = e[2] * e[0] * v - e[1] * s; R[7] = e[2] * e[1] * v + e[0] * s; R[8] = c + e[2] * e[2] * v;
}

/* world poses of the 22 geoms (same kinematic chain as the oracle's kinematics(): X = R_b x0 + t_b) */
static void geoms_world(const double* P, const double* qpos, Geom* g) {
    double R[NB][9], t[NB][3];
    const double* q = qpos + 3;
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]), w = q[0] / n, x = q[1] / n, y = q[2] / n, z = q[3] / n;
    double R0[9] = {w * w + x * x - y * y - z * z, 2 * (x * y - w * z), 2 * (x * z + w * y), 2 * (x * y + w * z), w * w - x * x + y * y - z * z, 2 * (y * z - w * x),
                    2 * (x * z - w * y), 2 * (y * z + w * x), w * w - x * x - y * y + z * z};
    memcpy(R[0], R0, sizeof R0);
    for (int i = 0; i < 3; i++) t[0][i] = qpos[i];
    for (int b = 1; b < NB; b++) {
        const double* H = P + JB_P_HINGE + (b - 1) * JB_HINGE_STRIDE;
        int p = PARENT_[b];
        double Rj[9], a_rot[3], tmp[3];
        rodrigues_(Rj, H + JB_H_AXIS, qpos[7 + (b - 1)]);
        mm3(R[b], R[p], Rj);
        mv3(a_rot, Rj, H + JB_H_ANCHOR);
        for (int i = 0; i < 3; i++) tmp[i] = H[JB_H_ANCHOR + i] - a_rot[i];
        mv3(t[b], R[p], tmp);
        for (int i = 0; i < 3; i++) t[b][i] += t[p][i];
    }
    for (int k = 0; k < JB_NGEOM; k++) {
        const double* G = P + JB_P_GEOM + k * JB_GEOM_STRIDE;
        int b = (int)G[JB_G_BODY];
        g[k].type = (int)G[JB_G_TYPE];
        mv3(g[k].c, R[b], G + JB_G_CENTER);
        for (int i = 0; i < 3; i++) { g[k].c[i] += t[b][i]; g[k].s[i] = G[JB_G_SIZE + i]; }
        mm3(g[k].R, R[b], G + JB_G_ROT);
        const double* s = g[k].s;
        g[k].rb = g[k].type == JB_GEOM_SPHERE ? s[0] : g[k].type == JB_GEOM_CYLINDER ? sqrt(s[0] * s[0] + s[1] * s[1])
                : g[k].type == JB_GEOM_BOX ? sqrt(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]) : fmax(s[0], fmax(s[1], s[2]));
    }
}

/* support point of a geom in world direction d (spheres are handled as points + radius by the caller) */
static void support(const Geom* g, const double* d, double* out) {
    double dl[3], sl[3] = {0, 0, 0};
    mtv3(dl, g->R, d);
    const double* s = g->s;
    if (g->type == JB_GEOM_CYLINDER) {
        double r = sqrt(dl[0] * dl[0] + dl[1] * dl[1]);
        if (r > 1e-300) { sl[0] = s[0] * dl[0] / r; sl[1] = s[0] * dl[1] / r; }
        sl[2] = dl[2] >= 0 ? s[1] : -s[1];
    } else if (g->type == JB_GEOM_BOX) {
        for (int i = 0; i < 3; i++) sl[i] = dl[i] >= 0 ? s[i] : -s[i];
    } else if (g->type == JB_GEOM_ELLIPSOID) {
        double den = sqrt(s[0] * s[0] * dl[0] * dl[0] + s[1] * s[1] * dl[1] * dl[1] + s[2] * s[2] * dl[2] * dl[2]);
        if (den > 1e-300) for (int i = 0; i < 3; i++) sl[i] = s[i] * s[i] * dl[i] / den;
    }
    mv3(out, g->R, sl);
    for (int i = 0; i < 3; i++) out[i] += g->c[i];
}

/* closest point of a simplex (1-4 vertices, rows of W) to the origin: barycentric weights in lam, the subset that carries
 * the point is compacted to the front; returns the new vertex count (0: the origin is inside a tetrahedron).
 * Exhaustive over the sub-simplices (<= 15), each solved by its Gram system: small, branch-light and robust in fp64. */
static int closest_simplex(double W[4][3], int n, double* v) {
    int best_mask = 0, best_n = 0;
    double best_d2 = INFINITY, best_lam[4] = {0, 0, 0, 0};
    for (int mask = 1; mask < (1 << n); mask++) {
        int idx[4], m = 0;
        for (int i = 0; i < n; i++) if (mask >> i & 1) idx[m++] = i;
        /* minimise |sum lam_i w_i|^2 with sum lam = 1 over the affine hull; accept when every lam_i >= 0 */
        double lam[4] = {1, 0, 0, 0};
        if (m > 1) {
            /* unknowns mu_k (k=1..m-1): p = w_0 + sum mu_k (w_k - w_0);  (E^T E) mu = -E^T w_0 */
            double E[3][3], G[3][3], r[3], mu[3] = {0, 0, 0};
            for (int k = 1; k < m; k++) for (int c = 0; c < 3; c++) E[k - 1][c] = W[idx[k]][c] - W[idx[0]][c];
            for (int a = 0; a < m - 1; a++) { r[a] = -dot_(E[a], W[idx[0]]); for (int b = 0; b < m - 1; b++) G[a][b] = dot_(E[a], E[b]); }
            int ok = 1;
            if (m == 2) { if (G[0][0] > 0) mu[0] = r[0] / G[0][0]; else ok = 0; }
            else if (m == 3) {
                double det = G[0][0] * G[1][1] - G[0][1] * G[1][0];
                if (fabs(det) > 1e-300) { mu[0] = (r[0] * G[1][1] - r[1] * G[0][1]) / det; mu[1] = (G[0][0] * r[1] - G[1][0] * r[0]) / det; } else ok = 0;
            } else {
                double det = G[0][0] * (G[1][1] * G[2][2] - G[1][2] * G[2][1]) - G[0][1] * (G[1][0] * G[2][2] - G[1][2] * G[2][0]) + G[0][2] * (G[1][0] * G[2][1] - G[1][1] * G[2][0]);
                if (fabs(det) > 1e-300) {
                    mu[0] = (r[0] * (G[1][1] * G[2][2] - G[1][2] * G[2][1]) - G[0][1] * (r[1] * G[2][2] - G[1][2] * r[2]) + G[0][2] * (r[1] * G[2][1] - G[1][1] * r[2])) / det;
                    mu[1] = (G[0][0] * (r[1] * G[2][2] - G[1][2] * r[2]) - r[0] * (G[1][0] * G[2][2] - G[1][2] * G[2][0]) + G[0][2] * (G[1][0] * r[2] - r[1] * G[2][0])) / det;
                    mu[2] = (G[0][0] * (G[1][1] * r[2] - r[1] * G[2][1]) - G[0][1] * (G[1][0] * r[2] - r[1] * G[2][0]) + r[0] * (G[1][0] * G[2][1] - G[1][1] * G[2][0])) / det;
                } else ok = 0;
            }
            if (!ok) continue;
            lam[0] = 1;
            for (int k = 1; k < m; k++) { lam[k] = mu[k - 1]; lam[0] -= mu[k - 1]; }
            int neg = 0;
            for (int k = 0; k < m; k++) if (lam[k] < -1e-14) neg = 1;
            if (neg) continue;
        }
        double p[3] = {0, 0, 0};
        for (int k = 0; k < m; k++) for (int c = 0; c < 3; c++) p[c] += lam[k] * W[idx[k]][c];
        double d2 = dot_(p, p);
        if (m == 4 && d2 > 1e-22) continue;      /* a (nearly flat) tetrahedron only counts when it really contains the origin */
        if (d2 < best_d2) { best_d2 = d2; best_mask = mask; best_n = m; memset(best_lam, 0, sizeof best_lam); for (int k = 0; k < m; k++) best_lam[k] = lam[k]; }
    }
    double Wn[4][3];
    int m = 0;
    v[0] = v[1] = v[2] = 0;
    for (int i = 0; i < n; i++) if (best_mask >> i & 1) { memcpy(Wn[m], W[i], sizeof Wn[m]); for (int c = 0; c < 3; c++) v[c] += best_lam[m] * W[i][c]; m++; }
    memcpy(W, Wn, sizeof(double) * 3 * m);
    if (best_n == 4) return 0;                    /* a proper tetrahedron carries the origin: the shapes intersect */
    return m;
}

/* distance between two convex geoms (0 when they intersect); spheres enter as points, their radii are subtracted */
static double gjk_distance(const Geom* a, const Geom* b) {
    double ra = a->type == JB_GEOM_SPHERE ? a->s[0] : 0.0, rb = b->type == JB_GEOM_SPHERE ? b->s[0] : 0.0;
    double W[4][3], v[3], sa[3], sb[3], w[3], mv[3];
    int n = 0;
    for (int i = 0; i < 3; i++) v[i] = a->c[i] - b->c[i];
    if (dot_(v, v) < 1e-30) { v[0] = 1; v[1] = v[2] = 0; }
    for (int it = 0; it < 200; it++) {
        for (int i = 0; i < 3; i++) mv[i] = -v[i];
        if (a->type == JB_GEOM_SPHERE) memcpy(sa, a->c, sizeof sa); else support(a, mv, sa);
        if (b->type == JB_GEOM_SPHERE) memcpy(sb, b->c, sizeof sb); else support(b, v, sb);
        for (int i = 0; i < 3; i++) w[i] = sa[i] - sb[i];
        double vv = dot_(v, v), vw = dot_(v, w);
        if (it > 0 && vv - vw <= 1e-12 * vv + 1e-24) break;          /* no further progress towards the origin */
        memcpy(W[n++], w, sizeof w);
        n = closest_simplex(W, n, v);
        if (n == 0 || dot_(v, v) < 1e-24) return 0.0;          /* the cores intersect */
    }
    double d = sqrt(dot_(v, v)) - ra - rb;
    return d > 0 ? d : 0.0;
}

/* exact distance of two posed geoms given as (type, centre, rotation, size): for the oracle's contact-activation margin of
 * its geom-geom candidates (jb_oracle.c) */
double jbo_posed_geom_distance(int ta, const double* ca, const double* Ra, const double* sa, int tb, const double* cb, const double* Rb, const double* sb) {
    Geom a, b;
    a.type = ta; memcpy(a.c, ca, sizeof a.c); memcpy(a.R, Ra, sizeof a.R); memcpy(a.s, sa, sizeof a.s); a.rb = 0;
    b.type = tb; memcpy(b.c, cb, sizeof b.c); memcpy(b.R, Rb, sizeof b.R); memcpy(b.s, sb, sizeof b.s); b.rb = 0;
    return gjk_distance(&a, &b);
}

static int pair_tested(int bi, int bj) { return bi != bj && PARENT_[bi] != bj && PARENT_[bj] != bi; }

/* minimum distance over all geom pairs MuJoCo's filters would let through; pair: the two geom indices (nullable) */
/* skip_simulated: leave out the pairs the simulator itself collides since round 3 - the mass ellipsoid (geom 21) against the four
 * upper-leg cylinders (geoms 4, 8, 12, 16) - so that the minimum is over the pairs it still ignores */
static double pair_clearance_core(const double* P, const double* qpos, int* pair, int skip_simulated) {
    Geom g[JB_NGEOM];
    geoms_world(P, qpos, g);
    double best = INFINITY;
    int bi = -1, bj = -1;
    for (int i = 0; i < JB_NGEOM; i++)
        for (int j = i + 1; j < JB_NGEOM; j++) {
            if (!pair_tested((int)P[JB_P_GEOM + i * JB_GEOM_STRIDE + JB_G_BODY], (int)P[JB_P_GEOM + j * JB_GEOM_STRIDE + JB_G_BODY])) continue;
            if (skip_simulated && (j == 21 || j == 20) && i >= 4 && i < 20 && ((i - 4) & 3) == 0) continue;      /* (round 5: the thread, geom 20, against the upper legs too) */
            double dc[3] = {g[i].c[0] - g[j].c[0], g[i].c[1] - g[j].c[1], g[i].c[2] - g[j].c[2]};
            if (sqrt(dot_(dc, dc)) - g[i].rb - g[j].rb >= best) continue;              /* bounding spheres cannot beat the minimum */
            double d = gjk_distance(&g[i], &g[j]);
            if (d < best) { best = d; bi = i; bj = j; }
        }
    if (pair) { pair[0] = bi; pair[1] = bj; }
    return best;
}
double jbo_pair_clearance(const double* P, const double* qpos, int* pair) { return pair_clearance_core(P, qpos, pair, 0); }
/* the same restricted to pairs with one geom on a body of mask_a and the other on a body of mask_b (bit b = body b) */
double jbo_pair_clearance_masked(const double* P, const double* qpos, unsigned mask_a, unsigned mask_b) {
    Geom g[JB_NGEOM];
    geoms_world(P, qpos, g);
    double best = INFINITY;
    for (int i = 0; i < JB_NGEOM; i++)
        for (int j = i + 1; j < JB_NGEOM; j++) {
            int bi = (int)P[JB_P_GEOM + i * JB_GEOM_STRIDE + JB_G_BODY], bj = (int)P[JB_P_GEOM + j * JB_GEOM_STRIDE + JB_G_BODY];
            if (!pair_tested(bi, bj)) continue;
            if (!(((mask_a >> bi & 1u) && (mask_b >> bj & 1u)) || ((mask_a >> bj & 1u) && (mask_b >> bi & 1u)))) continue;
            double dc[3] = {g[i].c[0] - g[j].c[0], g[i].c[1] - g[j].c[1], g[i].c[2] - g[j].c[2]};
            if (sqrt(dot_(dc, dc)) - g[i].rb - g[j].rb >= best) continue;
            double d = gjk_distance(&g[i], &g[j]);
            if (d < best) best = d;
        }
    return best;
}
/* rest pose, minimum over n_phi motor angles, of the mass-body <-> leg-body clearance: can the eccentric mass turn freely? */
void jbo_mass_sweep_clearance_batch(const double* P, int per_env_model, int n, int n_phi, double* out) {
#ifdef _OPENMP
#pragma omp parallel for schedule(static)
#endif
    for (int i = 0; i < n; i++) {
        const double* Pi = per_env_model ? P + (size_t)i * JB_NPARAM : P;
        double q[JB_NQ] = {0}, best = INFINITY;
        for (int k = 0; k < 3; k++) q[k] = Pi[JB_P_ROOTPOS0 + k];
        q[3] = 1.0;
        for (int a = 0; a < n_phi; a++) {
            q[15] = 2.0 * M_PI * a / n_phi;
            double d = jbo_pair_clearance_masked(Pi, q, 1u << 9, 0x1FEu);
            if (d < best) best = d;
        }
        out[i] = best;
    }
}
double jbo_geom_distance(const double* P, const double* qpos, int gi, int gj) {
    Geom g[JB_NGEOM];
    geoms_world(P, qpos, g);
    return gjk_distance(&g[gi], &g[gj]);
}
int jbo_num_tested_pairs(const double* P) {
    int n = 0;
    for (int i = 0; i < JB_NGEOM; i++) for (int j = i + 1; j < JB_NGEOM; j++)
        n += pair_tested((int)P[JB_P_GEOM + i * JB_GEOM_STRIDE + JB_G_BODY], (int)P[JB_P_GEOM + j * JB_GEOM_STRIDE + JB_G_BODY]);
    return n;
}
void jbo_pair_clearance_batch2(const double* P, int per_env_model, int n, const double* qpos, double* out, int* pairs, int nthreads, int skip_simulated) {
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel for schedule(static)
#endif
    for (int i = 0; i < n; i++)
        out[i] = pair_clearance_core(per_env_model ? P + (size_t)i * JB_NPARAM : P, qpos + (size_t)i * JB_NQ, pairs ? pairs + 2 * i : (int*)0, skip_simulated);
}
void jbo_pair_clearance_batch(const double* P, int per_env_model, int n, const double* qpos, double* out, int* pairs, int nthreads) {
    jbo_pair_clearance_batch2(P, per_env_model, n, qpos, out, pairs, nthreads, 0);
}
/* world pose of a geom, for the tests' independent brute-force check */
void jbo_geom_world(const double* P, const double* qpos, int gi, double* center, double* R, double* size) {
    Geom g[JB_NGEOM];
    geoms_world(P, qpos, g);
    memcpy(center, g[gi].c, sizeof g[gi].c); memcpy(R, g[gi].R, sizeof g[gi].R); memcpy(size, g[gi].s, sizeof g[gi].s);
}

// jb_model_build.hpp — compiled parameter table (include/jitterbug_model.h) -> per-lane constant table (jb_sim.hpp LM_*).
// Host-side, plain C++.  Used by the HIP library's host code and by the host test harness.
#pragma once
#include <cmath>

#include "../../include/jitterbug_model.h"
#include "jb_sim.hpp"

namespace jb {

// returns 0 on success, <0 when the table asks for something the kernel does not implement
template <typename T> JB_HD int build_lane_model(const double* P, int leg, T* out) {
    for (int i = 0; i < LM_COUNT; i++) out[i] = T(0);
    auto body = [&](int b) { return P + JB_P_BODY + b * JB_BODY_STRIDE; };
    auto hinge = [&](int h) { return P + JB_P_HINGE + h * JB_HINGE_STRIDE; };
    auto geom = [&](int g) { return P + JB_P_GEOM + g * JB_GEOM_STRIDE; };
    auto put3 = [&](int at, const double* v) { for (int i = 0; i < 3; i++) out[at + i] = T(v[i]); };
    auto put3d = [&](int at, const double* v, const double* ref) { for (int i = 0; i < 3; i++) out[at + i] = T(v[i] - ref[i]); };
    auto norm3d = [&](const double* v, const double* ref) { double s = 0; for (int i = 0; i < 3; i++) s += (v[i] - ref[i]) * (v[i] - ref[i]); return sqrt(s); };
    const double zero3[3] = {0, 0, 0};

    const double h = P[JB_P_TIMESTEP];
    out[LM_H] = T(h);
    put3(LM_GRAV, P + JB_P_GRAVITY);
    double tc = P[JB_P_SOLREF] < 2 * h ? 2 * h : P[JB_P_SOLREF], dr = P[JB_P_SOLREF + 1], dmax = P[JB_P_SOLIMP + 1];
    if (!(P[JB_P_SOLREF] > 0)) return -1;
    out[LM_KK] = T(1.0 / (dmax * dmax * tc * tc * dr * dr));
    out[LM_BB] = T(2.0 / (dmax * tc));
    out[LM_IMP_D0] = T(P[JB_P_SOLIMP]); out[LM_IMP_DW] = T(P[JB_P_SOLIMP + 1]); out[LM_IMP_IW] = T(1.0 / P[JB_P_SOLIMP + 2]);
    out[LM_IMP_MID] = T(P[JB_P_SOLIMP + 3]);
    if (!(P[JB_P_SOLIMP + 3] > 0.0 && P[JB_P_SOLIMP + 3] < 1.0)) return -6;      // midpoint must lie inside (0, 1)
    out[LM_IMP_IMID] = T(1.0 / P[JB_P_SOLIMP + 3]); out[LM_IMP_I1MID] = T(1.0 / (1.0 - P[JB_P_SOLIMP + 3]));
    if (P[JB_P_SOLIMP + 4] != 2.0) return -2;
    out[LM_MU] = T(P[JB_P_FRICTION] * sqrt(1.0 / P[JB_P_IMPRATIO]));
    out[LM_FR2] = T(P[JB_P_FRICTION] * P[JB_P_FRICTION]);
    out[LM_GEAR] = T(P[JB_P_GEAR]); out[LM_GAIN] = T(P[JB_P_GAIN]);
    put3(LM_BIAS, P + JB_P_BIASPRM);
    out[LM_CTRL_LO] = T(P[JB_P_CTRLRANGE]); out[LM_CTRL_HI] = T(P[JB_P_CTRLRANGE + 1]);
    double mtot = 0;
    for (int b = 0; b < JB_NBODY; b++) mtot += body(b)[JB_B_MASS];
    out[LM_MTOT] = T(mtot);
    out[LM_TARGET_Z] = T(P[JB_P_TARGETZ]); out[LM_ROOT_Z0] = T(P[JB_P_ROOTPOS0 + 2]);

    // root body
    out[LM_M0] = T(body(0)[JB_B_MASS]); put3(LM_C0, body(0) + JB_B_COM);
    for (int i = 0; i < 6; i++) out[LM_I0 + i] = T(body(0)[JB_B_INERTIA + i]);
    out[LM_TRAN0] = T(body(0)[JB_B_INVW_TRAN]);
    // motor body
    const double* am = hinge(8) + JB_H_ANCHOR;
    out[LM_MM] = T(body(9)[JB_B_MASS]); put3(LM_AM, am); put3(LM_EM, hinge(8) + JB_H_AXIS);
    put3d(LM_DCM, body(9) + JB_B_COM, am);
    for (int i = 0; i < 6; i++) out[LM_IM + i] = T(body(9)[JB_B_INERTIA + i]);
    out[LM_TRANM] = T(body(9)[JB_B_INVW_TRAN]);
    // own leg
    const int bu = 1 + 2 * leg, bl = 2 + 2 * leg, hs = 2 * leg, hk = 2 * leg + 1;
    const double* a1 = hinge(hs) + JB_H_ANCHOR;
    const double* a2 = hinge(hk) + JB_H_ANCHOR;
    put3(LM_A1, a1); put3(LM_E1, hinge(hs) + JB_H_AXIS); put3d(LM_DA2, a2, a1); put3(LM_E2, hinge(hk) + JB_H_AXIS);
    put3d(LM_DC1, body(bu) + JB_B_COM, a1); put3d(LM_DC2, body(bl) + JB_B_COM, a2);
    for (int i = 0; i < 6; i++) { out[LM_I1 + i] = T(body(bu)[JB_B_INERTIA + i]); out[LM_I2 + i] = T(body(bl)[JB_B_INERTIA + i]); }
    out[LM_M1] = T(body(bu)[JB_B_MASS]); out[LM_M2] = T(body(bl)[JB_B_MASS]);
    out[LM_K1] = T(hinge(hs)[JB_H_STIFFNESS]); out[LM_B1] = T(hinge(hs)[JB_H_DAMPING]);
    out[LM_K2] = T(hinge(hk)[JB_H_STIFFNESS]); out[LM_B2] = T(hinge(hk)[JB_H_DAMPING]);
    out[LM_TRAN1] = T(body(bu)[JB_B_INVW_TRAN]); out[LM_TRAN2] = T(body(bl)[JB_B_INVW_TRAN]);
    if (hinge(8)[JB_H_DAMPING] != 0.0 || hinge(8)[JB_H_STIFFNESS] != 0.0) return -3;   // motor hinge: no passive terms in the model
    const double *gu = geom(4 + 4 * leg), *gt = geom(5 + 4 * leg), *gl = geom(6 + 4 * leg), *gf = geom(7 + 4 * leg);
    if ((int)gu[JB_G_TYPE] != JB_GEOM_CYLINDER || (int)gt[JB_G_TYPE] != JB_GEOM_SPHERE || (int)gl[JB_G_TYPE] != JB_GEOM_CYLINDER || (int)gf[JB_G_TYPE] != JB_GEOM_SPHERE) return -4;
    auto put_cyl = [&](int at_d, int at_ax, int at_xa, int at_r, int at_h, const double* g, const double* ref) {
        put3d(at_d, g + JB_G_CENTER, ref);
        const double* Rg = g + JB_G_ROT;
        out[at_ax] = T(Rg[2]); out[at_ax + 1] = T(Rg[5]); out[at_ax + 2] = T(Rg[8]);
        out[at_xa] = T(Rg[0]); out[at_xa + 1] = T(Rg[3]); out[at_xa + 2] = T(Rg[6]);
        out[at_r] = T(g[JB_G_SIZE]); out[at_h] = T(g[JB_G_SIZE + 1]);
    };
    put3d(LM_DFOOT, gf + JB_G_CENTER, a2); out[LM_FOOT_R] = T(gf[JB_G_SIZE]);
    put_cyl(LM_LC_D, LM_LC_AX, LM_LC_XA, LM_LC_R, LM_LC_H, gl, a2);
    put_cyl(LM_UC_D, LM_UC_AX, LM_UC_XA, LM_UC_R, LM_UC_H, gu, a1);
    put3d(LM_DTIP, gt + JB_G_CENTER, a1); out[LM_TIP_R] = T(gt[JB_G_SIZE]);
    // broadphase sphere of the upper leg: centre = upper cylinder centre; the knee tip sits at the cylinder's far end
    put3(LM_BS_LEG_C, gu + JB_G_CENTER);
    {
        double rc = sqrt(gu[JB_G_SIZE] * gu[JB_G_SIZE] + gu[JB_G_SIZE + 1] * gu[JB_G_SIZE + 1]);
        double rt = norm3d(gt + JB_G_CENTER, gu + JB_G_CENTER) + gt[JB_G_SIZE];
        double slack = 0.3 * norm3d(gu + JB_G_CENTER, a1) + 1e-3;      // the centre moves with the shoulder angle (|th1| <= 0.3 here)
        out[LM_BS_LEG_R] = T((rc > rt ? rc : rt) + slack);
    }
    // lane-assigned root / motor-body geoms; disabled slots get finite, harmless geometry
    out[LM_XB_EN] = T(0); out[LM_XC_EN] = T(0); out[LM_XE_EN] = T(0); out[LM_X_ONM] = T(0);
    for (int i = 0; i < 9; i++) { out[LM_XB_R + i] = T(i % 4 == 0); out[LM_XE_R + i] = T(i % 4 == 0); }
    out[LM_XC_AX + 2] = T(1); out[LM_XC_XA] = T(1);
    out[LM_XE_S] = out[LM_XE_S + 1] = out[LM_XE_S + 2] = T(1);
    double xs_c[2][3], xs_r[2]; int xs_n = 0;
    auto put_box = [&](const double* g) {
        out[LM_XB_EN] = T(1); put3(LM_XB_C, g + JB_G_CENTER);
        for (int i = 0; i < 9; i++) out[LM_XB_R + i] = T(g[JB_G_ROT + i]);
        put3(LM_XB_S, g + JB_G_SIZE);
        for (int i = 0; i < 3; i++) xs_c[xs_n][i] = g[JB_G_CENTER + i];
        xs_r[xs_n++] = norm3d(g + JB_G_SIZE, zero3);
    };
    auto put_xcyl = [&](const double* g, const double* ref) {
        out[LM_XC_EN] = T(1);
        put_cyl(LM_XC_C, LM_XC_AX, LM_XC_XA, LM_XC_R, LM_XC_H, g, zero3);
        (void)ref;
        for (int i = 0; i < 3; i++) xs_c[xs_n][i] = g[JB_G_CENTER + i];
        xs_r[xs_n++] = sqrt(g[JB_G_SIZE] * g[JB_G_SIZE] + g[JB_G_SIZE + 1] * g[JB_G_SIZE + 1]);
    };
    auto put_ell = [&](const double* g, const double* ref) {
        out[LM_XE_EN] = T(1); put3(LM_XE_C, g + JB_G_CENTER);
        for (int i = 0; i < 9; i++) out[LM_XE_R + i] = T(g[JB_G_ROT + i]);
        put3(LM_XE_S, g + JB_G_SIZE);
        double mx = g[JB_G_SIZE];
        for (int i = 1; i < 3; i++) if (g[JB_G_SIZE + i] > mx) mx = g[JB_G_SIZE + i];
        (void)ref;
        for (int i = 0; i < 3; i++) xs_c[xs_n][i] = g[JB_G_CENTER + i];
        xs_r[xs_n++] = mx;
    };
    if (leg == 0) { if ((int)geom(0)[JB_G_TYPE] != JB_GEOM_BOX) return -5; put_box(geom(0)); }
    if (leg == 1) { if ((int)geom(1)[JB_G_TYPE] != JB_GEOM_BOX) return -5; put_box(geom(1)); }
    if (leg == 2) {
        if ((int)geom(2)[JB_G_TYPE] != JB_GEOM_CYLINDER || (int)geom(3)[JB_G_TYPE] != JB_GEOM_ELLIPSOID) return -5;
        put_xcyl(geom(2), zero3); put_ell(geom(3), zero3);
    }
    if (leg == 3) {
        if ((int)geom(20)[JB_G_TYPE] != JB_GEOM_CYLINDER || (int)geom(21)[JB_G_TYPE] != JB_GEOM_ELLIPSOID) return -5;
        put_xcyl(geom(20), am); put_ell(geom(21), am);
        out[LM_X_ONM] = T(1);
    }
    {   // the eccentric-mass ellipsoid for every lane (PAIR kernels: its contact with the lane's own upper-leg cylinder)
        const double* ge = geom(21);
        put3(LM_PE_C, ge + JB_G_CENTER);
        for (int i = 0; i < 9; i++) out[LM_PE_R + i] = T(ge[JB_G_ROT + i]);
        put3(LM_PE_S, ge + JB_G_SIZE);
        // (no ellipsoid: x <= 0 switches the pair off; the y entry stays POSITIVE - its sign is the thread flag's, below, and must not double as this sentinel)
        for (int i = 0; i < 3; i++) out[LM_PE_IS + i] = T((int)ge[JB_G_TYPE] == JB_GEOM_ELLIPSOID ? 1.0 / (ge[JB_G_SIZE + i] + gu[JB_G_SIZE] + 2e-4) : (i == 1 ? 1.0 : -1.0));
    }
    {   // the motor-axis thread for every lane (PAIR kernels: its contact with the lane's own upper-leg cylinder), and the flag that says whether
        // this lane's leg can come near it at all: the smallest distance of the two AXES over the shoulder angles |th1| <= 0.4 rad (rollouts
        // stay below 0.15) and a full turn of the motor, against r_thread + r_leg + 1 mm.  The flag is the SIGN of LM_PE_IS + 1 (negative = near; the magnitude is the mass pair's scale, or 1 without an ellipsoid).
        // Outside that envelope of shoulder angles the kernel does not trust the flag: it runs the broad phase (substep_impl).
        const double* gth = geom(20);
        put3(LM_PT_C, gth + JB_G_CENTER);
        const double tax[3] = {gth[JB_G_ROT + 2], gth[JB_G_ROT + 5], gth[JB_G_ROT + 8]};
        for (int i = 0; i < 3; i++) out[LM_PT_AX + i] = T(tax[i]);
        out[LM_PT_R] = T(gth[JB_G_SIZE]); out[LM_PT_H] = T(gth[JB_G_SIZE + 1]);
        bool near = false;
        if ((int)gth[JB_G_TYPE] == JB_GEOM_CYLINDER && (int)gu[JB_G_TYPE] == JB_GEOM_CYLINDER) {
            const double* h1 = hinge(2 * leg);
            const double* hm = hinge(8);
            const double* a1 = h1 + JB_H_ANCHOR;
            const double* e1 = h1 + JB_H_AXIS;
            const double* am_ = hm + JB_H_ANCHOR;
            const double* em = hm + JB_H_AXIS;
            const double uax[3] = {gu[JB_G_ROT + 2], gu[JB_G_ROT + 5], gu[JB_G_ROT + 8]};
            auto rot = [](const double* e, double ang, const double* v, double* o) {      // Rodrigues
                const double c = cos(ang), s = sin(ang), d = e[0] * v[0] + e[1] * v[1] + e[2] * v[2];
                const double cr[3] = {e[1] * v[2] - e[2] * v[1], e[2] * v[0] - e[0] * v[2], e[0] * v[1] - e[1] * v[0]};
                for (int i = 0; i < 3; i++) o[i] = v[i] * c + cr[i] * s + e[i] * d * (1 - c);
            };
            const double lim = gth[JB_G_SIZE] + gu[JB_G_SIZE] + 1e-3;
            for (int ia = 0; ia <= 8 && !near; ia++) {
                const double th = -0.4 + 0.1 * ia;
                double rel[3], uc[3], ua[3];
                for (int i = 0; i < 3; i++) rel[i] = gu[JB_G_CENTER + i] - a1[i];
                rot(e1, th, rel, uc);
                for (int i = 0; i < 3; i++) uc[i] += a1[i];
                rot(e1, th, uax, ua);
                for (int ip = 0; ip < 16 && !near; ip++) {
                    const double ph = 0.39269908169872414 * ip;
                    double relt[3], tc[3], ta[3];
                    for (int i = 0; i < 3; i++) relt[i] = gth[JB_G_CENTER + i] - am_[i];
                    rot(em, ph, relt, tc);
                    for (int i = 0; i < 3; i++) tc[i] += am_[i];
                    rot(em, ph, tax, ta);
                    // segment - segment distance (Ericson 5.1.9)
                    const double r[3] = {uc[0] - tc[0], uc[1] - tc[1], uc[2] - tc[2]};
                    const double b = ua[0] * ta[0] + ua[1] * ta[1] + ua[2] * ta[2], c = ua[0] * r[0] + ua[1] * r[1] + ua[2] * r[2], f = ta[0] * r[0] + ta[1] * r[1] + ta[2] * r[2];
                    const double den = 1.0 - b * b, hu = gu[JB_G_SIZE + 1], ht = gth[JB_G_SIZE + 1];
                    double sq = den > 1e-12 ? (b * f - c) / den : 0.0;
                    sq = sq < -hu ? -hu : (sq > hu ? hu : sq);
                    double tq = b * sq + f;
                    if (tq < -ht || tq > ht) { tq = tq < -ht ? -ht : ht; sq = b * tq - c; sq = sq < -hu ? -hu : (sq > hu ? hu : sq); }
                    double d2 = 0;
                    for (int i = 0; i < 3; i++) { const double dd = r[i] + sq * ua[i] - tq * ta[i]; d2 += dd * dd; }
                    if (d2 < lim * lim) near = true;
                }
            }
        }
        out[LM_PE_IS + 1] = near ? -std::fabs(out[LM_PE_IS + 1]) : std::fabs(out[LM_PE_IS + 1]);
    }
    {   // broadphase boxes (LM_BX): oriented boxes around the lane's root-body geoms; motor-body geoms (lane 3) get one
        // axis-aligned cube centred on the motor axis (invariant under the motor angle).  Unused boxes have negative sizes.
        for (int k = 0; k < 2; k++) {
            T* bx = out + LM_BX + 15 * k;
            for (int i = 0; i < 15; i++) bx[i] = T(0);
            bx[3] = bx[7] = bx[11] = T(1);
            bx[12] = bx[13] = bx[14] = T(-1);
        }
        const double margin = 2e-4;
        auto put_bx = [&](int k, const double* g, const double* hs) {
            T* bx = out + LM_BX + 15 * k;
            for (int i = 0; i < 3; i++) bx[i] = T(g[JB_G_CENTER + i]);
            const double* Rg = g + JB_G_ROT;
            for (int ax = 0; ax < 3; ax++) for (int i = 0; i < 3; i++) bx[3 + 3 * ax + i] = T(Rg[3 * i + ax]);
            for (int i = 0; i < 3; i++) bx[12 + i] = T(hs[i] + margin);
        };
        auto cyl_hs = [&](const double* g, double* hs) { hs[0] = hs[1] = g[JB_G_SIZE]; hs[2] = g[JB_G_SIZE + 1]; };
        double hs[3];
        if (leg == 0) put_bx(0, geom(0), geom(0) + JB_G_SIZE);
        if (leg == 1) put_bx(0, geom(1), geom(1) + JB_G_SIZE);
        if (leg == 2) { cyl_hs(geom(2), hs); put_bx(0, geom(2), hs); put_bx(1, geom(3), geom(3) + JB_G_SIZE); }
        if (leg == 3) {
            double c[3] = {0, 0, 0};
            for (int k = 0; k < xs_n; k++) for (int i = 0; i < 3; i++) c[i] += xs_c[k][i] / xs_n;
            const double* em = hinge(8) + JB_H_AXIS;
            double t = 0;
            for (int i = 0; i < 3; i++) t += (c[i] - am[i]) * em[i];
            for (int i = 0; i < 3; i++) c[i] = am[i] + t * em[i];
            double r = 0;
            for (int k = 0; k < xs_n; k++) { double d = norm3d(xs_c[k], c) + xs_r[k]; if (d > r) r = d; }
            T* bx = out + LM_BX;
            for (int i = 0; i < 3; i++) { bx[i] = T(c[i]); bx[12 + i] = T(r + margin); }
        }
        // first-level test: one sphere around the lane's boxes (centre = mean of the box centres, radius reaches every corner)
        double sc[3] = {0, 0, 0}, sr = 0;
        int nb = 0;
        for (int k = 0; k < 2; k++) if (out[LM_BX + 15 * k + 12] > T(0)) { for (int i = 0; i < 3; i++) sc[i] += (double)out[LM_BX + 15 * k + i]; nb++; }
        for (int i = 0; i < 3; i++) sc[i] = nb ? sc[i] / nb : 0.0;
        for (int k = 0; k < 2; k++) {
            const T* bx = out + LM_BX + 15 * k;
            if (!(bx[12] > T(0))) continue;
            double dc = 0, dh = 0;
            for (int i = 0; i < 3; i++) { dc += ((double)bx[i] - sc[i]) * ((double)bx[i] - sc[i]); dh += (double)bx[12 + i] * (double)bx[12 + i]; }
            double rr = sqrt(dc) + sqrt(dh);
            if (rr > sr) sr = rr;
        }
        for (int i = 0; i < 3; i++) out[LM_BS_BODY_C + i] = T(sc[i]);
        out[LM_BS_BODY_R] = T(nb ? sr + margin : -1.0);
    }
    return 0;
}

// the packed table of one env (LM_TABLE entries: jb_sim.hpp LM_INV): <0 when unsupported, -50 if an entry that is stored
// once differs between the legs (a bug in the table layout, not in the model)
template <typename T> JB_HD int build_packed_model(const double* P, T* out) {
    T tmp[4][LM_COUNT];
    for (int leg = 0; leg < 4; leg++) { int rc = build_lane_model<T>(P, leg, tmp[leg]); if (rc) return rc; }
    for (int i = 0; i < LM_INV; i++) {
        for (int leg = 1; leg < 4; leg++) if (!(tmp[leg][i] == tmp[0][i])) return -50;
        out[i] = tmp[0][i];
    }
    for (int i = LM_INV; i < LM_COUNT; i++) for (int leg = 0; leg < 4; leg++) out[lm_offset(i, leg)] = tmp[leg][i];
    return 0;
}

}  // namespace jb

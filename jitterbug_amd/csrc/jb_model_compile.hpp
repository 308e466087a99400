// jb_model_compile.hpp — native model compiler and domain randomiser, written once for the device and the host.
//
// What it replaces, per environment:
//   * the reference's domain randomisation, augment_Jitterbug (reference jitterbug_dmc/augmented_jitterbug.py:95-267): Gaussian
//     offsets of the leg ends and the motor axis, optional density / gear perturbations, hinge axes recomputed from the
//     perturbed geometry (:182-186, :212-215) — `apply_offsets`;
//   * the part of MuJoCo's MJCF compilation the model relies on (third party; restated in jitterbug_amd/model.py, which stays
//     the golden-checked definition): geom mass / inertia from density x solid primitive, fromto cylinders, body mass / COM /
//     inertia by the parallel-axis theorem, body_invweight0 = trace(J M^-1 J^T)/3 at qpos0 — `compile_model`;
//   * a validity check the reference does not have: can the eccentric mass turn without hitting a leg (`mass_sweep_clear`)?
//     The reference's own sigma (1.5 mm on the motor axis against 2.8 mm of nominal clearance) produces such robots in 3.6 % of
//     the draws; MuJoCo would simulate a mass-leg contact there, this simulator collides with the floor only.
// Output: the flat parameter table of include/jitterbug_model.h (JB_NPARAM doubles), which build_packed_model()
// (jb_model_build.hpp) turns into the kernel's lane constant table.  One thread per environment on the device
// (jb_randomise_kernel, jb_api.hip); fp64 throughout.
#pragma once
#include <cmath>
#include <cstdint>

#include "jb_lane.hpp"
#include "jb_model_compile_types.h"

namespace jb {

// offsets of one randomised model, in the reference's draw order (augmented_jitterbug.py:144-247)
enum AugOff : int {
    AO_GLOBAL_DENSITY = 0, AO_CORE1_DENSITY = 1, AO_CORE2_DENSITY = 2,
    AO_LEG = 3 /* 4 legs x [du(3) upper far end, dl(3) foot end] in XML order leg2, leg3, leg1, leg4 */,
    AO_MASS = AO_LEG + 24 /*3: motor axis / thread offset, already clipped*/, AO_GEAR = AO_MASS + 3, AO_COUNT = AO_GEAR + 1
};
enum AugFlags : int { AUG_LEGS = 1, AUG_MASS = 2, AUG_CORE1 = 4, AUG_CORE2 = 8, AUG_GLOBAL_DENSITY = 16, AUG_GEAR = 32 };

namespace mc {
JB_HD void cross(const double* a, const double* b, double* o) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
JB_HD double dot(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
JB_HD void unit(double* v) { double n = sqrt(dot(v, v)); v[0] /= n; v[1] /= n; v[2] /= n; }
JB_HD void matmul(const double* A, const double* B, double* C) {
    double t[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    for (int i = 0; i < 9; i++) C[i] = t[i];
}
// rotation taking the z axis onto v by the minimal rotation (MuJoCo's rule for fromto geoms; model.py _quat_z2vec)
JB_HD void z2vec(const double* vec, double* R) {
    for (int i = 0; i < 9; i++) R[i] = (i % 4 == 0) ? 1.0 : 0.0;
    double n = sqrt(dot(vec, vec));
    if (n < 1e-15) return;
    double v[3] = {vec[0] / n, vec[1] / n, vec[2] / n};
    double ax[3] = {-v[1], v[0], 0.0};          // z x v
    double s = sqrt(ax[0] * ax[0] + ax[1] * ax[1]);
    if (s < 1e-15) { if (v[2] < 0) { R[4] = -1.0; R[8] = -1.0; } return; }
    ax[0] /= s; ax[1] /= s;
    double ang = atan2(s, v[2]), sn = sin(ang), cs = 1.0 - cos(ang);
    double K[9] = {0, -ax[2], ax[1], ax[2], 0, -ax[0], -ax[1], ax[0], 0}, KK[9];
    matmul(K, K, KK);
    for (int i = 0; i < 9; i++) R[i] += sn * K[i] + cs * KK[i];
}
}  // namespace mc

// the reference's perturbations applied to a copy of the nominal spec (augment_spec in jitterbug_amd/augmented_jitterbug.py
// is the golden-checked definition of the same thing; off: AO_COUNT values)
JB_HD void apply_offsets(const JbNominalSpec& S0, int flags, const double* off, JbNominalSpec& S) {
    S = S0;
    if (flags & AUG_GLOBAL_DENSITY) S.default_density += off[AO_GLOBAL_DENSITY];                      // :144-148
    if (flags & AUG_CORE1) S.geoms[0].density += off[AO_CORE1_DENSITY];                              // :152-160
    if (flags & AUG_CORE2) S.geoms[1].density += off[AO_CORE2_DENSITY];
    if (flags & AUG_LEGS) {
        for (int l = 0; l < JB_NLEG; l++) {                                                          // :162-215
            const double* du = off + AO_LEG + 6 * l;
            const double* dl = du + 3;
            JbGeomSpec &uc = S.geoms[4 + 4 * l], &ut = S.geoms[5 + 4 * l], &lc = S.geoms[6 + 4 * l], &lf = S.geoms[7 + 4 * l];
            JbHingeSpec &hs = S.hinges[2 * l], &hk = S.hinges[2 * l + 1];
            for (int i = 0; i < 3; i++) { uc.fromto[i] += du[i]; ut.pos[i] += du[i]; }              // far (knee) end of the upper leg + its tip
            double vu[3] = {uc.fromto[3] - uc.fromto[0], uc.fromto[4] - uc.fromto[1], uc.fromto[5] - uc.fromto[2]};
            mc::unit(vu);
            const double ez[3] = {0, 0, 1};
            mc::cross(vu, ez, hs.axis); mc::unit(hs.axis);                                           // shoulder axis :182-186
            for (int i = 0; i < 3; i++) { lc.fromto[i] += dl[i]; lc.fromto[3 + i] += du[i]; lf.pos[i] += dl[i]; hk.pos[i] += du[i]; }
            double vl[3] = {lc.fromto[3] - lc.fromto[0], lc.fromto[4] - lc.fromto[1], lc.fromto[5] - lc.fromto[2]};
            mc::unit(vl);
            mc::cross(vu, vl, hk.axis); mc::unit(hk.axis);                                           // knee axis :212-215
        }
    }
    if (flags & AUG_MASS) {                                                                          // :217-241
        const double* dp = off + AO_MASS;
        JbGeomSpec& th = S.geoms[20];
        th.fromto[0] += dp[0]; th.fromto[1] += dp[1]; th.fromto[3] += dp[0]; th.fromto[4] += dp[1]; th.fromto[5] += dp[2];   // :224
        for (int i = 0; i < 3; i++) S.hinges[8].pos[i] += dp[i];                                     // :240-241
    }
    if (flags & AUG_GEAR) S.gear += off[AO_GEAR];                                                    // :243-247
}

// spec -> flat parameter table (model.py compile_spec restated); returns 0, or <0 if the mass matrix at qpos0 is not SPD
JB_HD int compile_model(const JbNominalSpec& S, double* P) {
    const int PARENT[JB_NBODY] = {-1, 0, 1, 0, 3, 0, 5, 0, 7, 0};
    for (int i = 0; i < JB_NPARAM; i++) P[i] = 0.0;
    P[JB_P_TIMESTEP] = S.timestep;
    for (int i = 0; i < 3; i++) { P[JB_P_GRAVITY + i] = S.gravity[i]; P[JB_P_BIASPRM + i] = S.biasprm[i]; P[JB_P_ROOTPOS0 + i] = S.root_pos[i]; }
    for (int i = 0; i < 2; i++) { P[JB_P_SOLREF + i] = S.solref[i]; P[JB_P_CTRLRANGE + i] = S.ctrlrange[i]; }
    for (int i = 0; i < 5; i++) P[JB_P_SOLIMP + i] = S.solimp[i];
    P[JB_P_FRICTION] = S.friction; P[JB_P_IMPRATIO] = S.impratio; P[JB_P_GEAR] = S.gear; P[JB_P_GAIN] = S.gain; P[JB_P_TARGETZ] = S.target_z;
    const double PI = 3.141592653589793;
    double bm[JB_NBODY], bc[JB_NBODY][3], bI[JB_NBODY][9];
    for (int b = 0; b < JB_NBODY; b++) { bm[b] = 0; for (int i = 0; i < 3; i++) bc[b][i] = 0; for (int i = 0; i < 9; i++) bI[b][i] = 0; }
    // pass 1: geoms (centre, frame, size, mass), body mass and COM
    double gm[JB_NGEOM];
    for (int g = 0; g < JB_NGEOM; g++) {
        const JbGeomSpec& G = S.geoms[g];
        double* o = P + JB_P_GEOM + g * JB_GEOM_STRIDE;
        double c[3], R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, sz[3] = {G.size[0], G.size[1], G.size[2]};
        if (G.has_fromto) {
            double d[3];
            for (int i = 0; i < 3; i++) { c[i] = 0.5 * (G.fromto[i] + G.fromto[3 + i]); d[i] = G.fromto[i] - G.fromto[3 + i]; }
            mc::z2vec(d, R);
            sz[1] = 0.5 * sqrt(mc::dot(d, d)); sz[2] = 0.0;
        } else for (int i = 0; i < 3; i++) c[i] = G.pos[i];
        const double rho = G.density >= 0 ? G.density : S.default_density;
        double vol = 0;
        if (G.type == JB_GEOM_SPHERE) { vol = 4.0 / 3.0 * PI * sz[0] * sz[0] * sz[0]; sz[1] = sz[2] = 0.0; }
        else if (G.type == JB_GEOM_CYLINDER) vol = PI * sz[0] * sz[0] * 2 * sz[1];
        else if (G.type == JB_GEOM_BOX) vol = 8 * sz[0] * sz[1] * sz[2];
        else vol = 4.0 / 3.0 * PI * sz[0] * sz[1] * sz[2];
        gm[g] = rho * vol;
        o[JB_G_TYPE] = G.type; o[JB_G_BODY] = G.body;
        for (int i = 0; i < 3; i++) { o[JB_G_CENTER + i] = c[i] - S.root_pos[i]; o[JB_G_SIZE + i] = sz[i]; }
        for (int i = 0; i < 9; i++) o[JB_G_ROT + i] = R[i];
        bm[G.body] += gm[g];
        for (int i = 0; i < 3; i++) bc[G.body][i] += gm[g] * c[i];
    }
    for (int b = 0; b < JB_NBODY; b++) for (int i = 0; i < 3; i++) bc[b][i] /= bm[b];
    // pass 2: body inertia about its COM (parallel axes), world axes
    for (int g = 0; g < JB_NGEOM; g++) {
        const JbGeomSpec& G = S.geoms[g];
        const double* o = P + JB_P_GEOM + g * JB_GEOM_STRIDE;
        const double *sz = o + JB_G_SIZE, *R = o + JB_G_ROT, m = gm[g];
        double Il[3];
        if (G.type == JB_GEOM_SPHERE) Il[0] = Il[1] = Il[2] = 0.4 * m * sz[0] * sz[0];
        else if (G.type == JB_GEOM_CYLINDER) { Il[0] = Il[1] = m * (3 * sz[0] * sz[0] + 4 * sz[1] * sz[1]) / 12.0; Il[2] = 0.5 * m * sz[0] * sz[0]; }
        else if (G.type == JB_GEOM_BOX) { Il[0] = m / 3 * (sz[1] * sz[1] + sz[2] * sz[2]); Il[1] = m / 3 * (sz[0] * sz[0] + sz[2] * sz[2]); Il[2] = m / 3 * (sz[0] * sz[0] + sz[1] * sz[1]); }
        else { Il[0] = m / 5 * (sz[1] * sz[1] + sz[2] * sz[2]); Il[1] = m / 5 * (sz[0] * sz[0] + sz[2] * sz[2]); Il[2] = m / 5 * (sz[0] * sz[0] + sz[1] * sz[1]); }
        double d[3];
        for (int i = 0; i < 3; i++) d[i] = (o[JB_G_CENTER + i] + S.root_pos[i]) - bc[G.body][i];
        const double dd = mc::dot(d, d);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) {
            double rir = 0;
            for (int k = 0; k < 3; k++) rir += R[3 * i + k] * Il[k] * R[3 * j + k];
            bI[G.body][3 * i + j] += rir + m * ((i == j ? dd : 0.0) - d[i] * d[j]);
        }
    }
    for (int b = 0; b < JB_NBODY; b++) {
        double* o = P + JB_P_BODY + b * JB_BODY_STRIDE;
        o[JB_B_MASS] = bm[b];
        for (int i = 0; i < 3; i++) o[JB_B_COM + i] = bc[b][i] - S.root_pos[i];
        const double* I = bI[b];
        o[JB_B_INERTIA + 0] = I[0]; o[JB_B_INERTIA + 1] = I[4]; o[JB_B_INERTIA + 2] = I[8]; o[JB_B_INERTIA + 3] = I[1]; o[JB_B_INERTIA + 4] = I[2]; o[JB_B_INERTIA + 5] = I[5];
    }
    for (int h = 0; h < JB_NHINGE; h++) {
        double* o = P + JB_P_HINGE + h * JB_HINGE_STRIDE;
        double ax[3] = {S.hinges[h].axis[0], S.hinges[h].axis[1], S.hinges[h].axis[2]};
        mc::unit(ax);
        for (int i = 0; i < 3; i++) { o[JB_H_ANCHOR + i] = S.hinges[h].pos[i] - S.root_pos[i]; o[JB_H_AXIS + i] = ax[i]; }
        o[JB_H_STIFFNESS] = S.hinges[h].stiffness; o[JB_H_DAMPING] = S.hinges[h].damping;
    }
    // body_invweight0: M at qpos0 by the Jacobian sum (model.py mass_matrix_qpos0), Cholesky, then trace(J M^-1 J^T)/3 per body.
    // Columns of a body's Jacobians: 3 root translations, 3 root rotations, the hinges on its path to the root (at most two).
    double M[JB_NV * JB_NV];
    for (int i = 0; i < JB_NV * JB_NV; i++) M[i] = 0.0;
    for (int b = 0; b < JB_NBODY; b++) {
        int dof[8], nd = 6;
        double Jv[8][3], Jw[8][3];
        const double *c = P + JB_P_BODY + b * JB_BODY_STRIDE + JB_B_COM, *ii = P + JB_P_BODY + b * JB_BODY_STRIDE + JB_B_INERTIA;
        const double I[9] = {ii[0], ii[3], ii[4], ii[3], ii[1], ii[5], ii[4], ii[5], ii[2]};
        for (int j = 0; j < 3; j++) {
            dof[j] = j; dof[3 + j] = 3 + j;
            for (int i = 0; i < 3; i++) { Jv[j][i] = (i == j); Jw[j][i] = 0.0; Jw[3 + j][i] = (i == j); }
            double e[3] = {double(j == 0), double(j == 1), double(j == 2)};
            mc::cross(e, c, Jv[3 + j]);
        }
        for (int a = b; a > 0; a = PARENT[a]) {
            const double* H = P + JB_P_HINGE + (a - 1) * JB_HINGE_STRIDE;
            double r[3] = {c[0] - H[JB_H_ANCHOR], c[1] - H[JB_H_ANCHOR + 1], c[2] - H[JB_H_ANCHOR + 2]};
            dof[nd] = 5 + a;
            for (int i = 0; i < 3; i++) Jw[nd][i] = H[JB_H_AXIS + i];
            mc::cross(H + JB_H_AXIS, r, Jv[nd]);
            nd++;
        }
        for (int x = 0; x < nd; x++) {
            double IJ[3];
            for (int i = 0; i < 3; i++) IJ[i] = I[3 * i] * Jw[x][0] + I[3 * i + 1] * Jw[x][1] + I[3 * i + 2] * Jw[x][2];
            for (int y = 0; y < nd; y++) M[dof[x] * JB_NV + dof[y]] += bm[b] * mc::dot(Jv[x], Jv[y]) + mc::dot(IJ, Jw[y]);
        }
    }
    double L[JB_NV * JB_NV];
    for (int i = 0; i < JB_NV * JB_NV; i++) L[i] = M[i];
    for (int j = 0; j < JB_NV; j++) {
        double s = L[j * JB_NV + j];
        for (int k = 0; k < j; k++) s -= L[j * JB_NV + k] * L[j * JB_NV + k];
        if (!(s > 0)) return -20;
        const double d = sqrt(s);
        L[j * JB_NV + j] = d;
        for (int i = j + 1; i < JB_NV; i++) {
            double t = L[i * JB_NV + j];
            for (int k = 0; k < j; k++) t -= L[i * JB_NV + k] * L[j * JB_NV + k];
            L[i * JB_NV + j] = t / d;
        }
    }
    for (int b = 0; b < JB_NBODY; b++) {
        // rows of Jv / Jw as dense NV vectors, solved one at a time:  trace(J M^-1 J^T) = sum_rows |L^-1 J_row^T|^2
        const double* c = P + JB_P_BODY + b * JB_BODY_STRIDE + JB_B_COM;
        double tv = 0, tw = 0;
        for (int row = 0; row < 3; row++) {
            for (int which = 0; which < 2; which++) {
                double x[JB_NV];
                for (int i = 0; i < JB_NV; i++) x[i] = 0.0;
                for (int j = 0; j < 3; j++) {
                    double e[3] = {double(j == 0), double(j == 1), double(j == 2)}, cr[3];
                    mc::cross(e, c, cr);
                    if (which == 0) { x[j] = (j == row); x[3 + j] = cr[row]; } else x[3 + j] = (j == row);
                }
                for (int a = b; a > 0; a = PARENT[a]) {
                    const double* H = P + JB_P_HINGE + (a - 1) * JB_HINGE_STRIDE;
                    double r[3] = {c[0] - H[JB_H_ANCHOR], c[1] - H[JB_H_ANCHOR + 1], c[2] - H[JB_H_ANCHOR + 2]}, cr[3];
                    mc::cross(H + JB_H_AXIS, r, cr);
                    x[5 + a] = which == 0 ? cr[row] : H[JB_H_AXIS + row];
                }
                double acc = 0;
                for (int i = 0; i < JB_NV; i++) {          // forward substitution L y = x, accumulate |y|^2
                    double s = x[i];
                    for (int k = 0; k < i; k++) s -= L[i * JB_NV + k] * x[k];
                    x[i] = s / L[i * JB_NV + i];
                    acc += x[i] * x[i];
                }
                if (which == 0) tv += acc; else tw += acc;
            }
        }
        P[JB_P_BODY + b * JB_BODY_STRIDE + JB_B_INVW_TRAN] = tv / 3.0;
        P[JB_P_BODY + b * JB_BODY_STRIDE + JB_B_INVW_ROT] = tw / 3.0;
    }
    return 0;
}

// ---- draws.  Philox4x32-10 keyed (seed, global env, attempt, block) -> Box-Muller normals; the reference draws from numpy's
// global RNG, so only the DISTRIBUTIONS are the reference's: N(0, sd) per component, the motor offset's
// y and z clipped below at -1 mm (np.clip(..., -0.001, 1.), augmented_jitterbug.py:219-221).
struct AugSigmas { double legs[3], mass_pos[3], core1_density, core2_density, global_density, gear; };
JB_HD void philox_raw(uint64_t seed, uint64_t env, uint32_t c2, uint32_t c3, uint32_t (&out)[4]) {
    uint32_t c0 = (uint32_t)env, c1 = (uint32_t)(env >> 32), k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
    for (int r = 0; r < 10; r++) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c0, p1 = (uint64_t)0xCD9E8D57u * c2;
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0, n1 = (uint32_t)p1, n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1, n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
JB_HD void draw_offsets(uint64_t seed, uint64_t env, uint32_t attempt, int flags, const AugSigmas& sd, double* off) {
    double nrm[32];
    for (int b = 0; b < 8; b++) {
        uint32_t r[4];
        philox_raw(seed ^ 0x6A6974746572ull /* "jitter" */, env, attempt, (uint32_t)b, r);
        for (int h = 0; h < 2; h++) {
            const double u0 = ((double)r[2 * h] + 0.5) * (1.0 / 4294967296.0), u1 = ((double)r[2 * h + 1] + 0.5) * (1.0 / 4294967296.0);
            const double rad = sqrt(-2.0 * log(u0)), ang = 6.283185307179586 * u1;
            nrm[4 * b + 2 * h] = rad * cos(ang); nrm[4 * b + 2 * h + 1] = rad * sin(ang);
        }
    }
    for (int i = 0; i < AO_COUNT; i++) off[i] = 0.0;
    int k = 0;
    const double g = nrm[k++], c1 = nrm[k++], c2 = nrm[k++];
    if (flags & AUG_GLOBAL_DENSITY) off[AO_GLOBAL_DENSITY] = g * sd.global_density;
    if (flags & AUG_CORE1) off[AO_CORE1_DENSITY] = c1 * sd.core1_density;
    if (flags & AUG_CORE2) off[AO_CORE2_DENSITY] = c2 * sd.core2_density;
    for (int l = 0; l < JB_NLEG; l++) for (int j = 0; j < 6; j++) { const double n = nrm[k++]; if (flags & AUG_LEGS) off[AO_LEG + 6 * l + j] = n * sd.legs[j % 3]; }
    for (int j = 0; j < 3; j++) {
        double v = nrm[k++] * sd.mass_pos[j];
        if (j > 0) v = v < -0.001 ? -0.001 : (v > 1.0 ? 1.0 : v);
        if (flags & AUG_MASS) off[AO_MASS + j] = v;
    }
    const double gr = nrm[k++];
    if (flags & AUG_GEAR) off[AO_GEAR] = gr * sd.gear;
}

// Can the eccentric mass turn freely?  Rest pose; every leg geom (cylinders as 33 points along their axis, spheres as their
// centre, each with its radius) against the thread cylinder and against the mass ellipsoid swept over 96 motor angles; the
// ellipsoid is inflated by the leg radius + margin (+0.25 mm: an ellipsoid with enlarged semi-axes under-covers the true offset
// body by up to ~0.15 mm at these sizes).  Conservative by sampling density only; tests/test_randomise.py holds it against the oracle's exact GJK sweep.
JB_HD bool mass_sweep_clear(const double* P, double margin) {
    const double* Hm = P + JB_P_HINGE + 8 * JB_HINGE_STRIDE;
    const double *am = Hm + JB_H_ANCHOR, *em = Hm + JB_H_AXIS;
    const double *Gt = P + JB_P_GEOM + 20 * JB_GEOM_STRIDE, *Ge = P + JB_P_GEOM + 21 * JB_GEOM_STRIDE;
    const double t_ax[3] = {Gt[JB_G_ROT + 2], Gt[JB_G_ROT + 5], Gt[JB_G_ROT + 8]};
    for (int g = 4; g < 20; g++) {
        const double* G = P + JB_P_GEOM + g * JB_GEOM_STRIDE;
        const bool cyl = (int)G[JB_G_TYPE] == JB_GEOM_CYLINDER;
        const double rad = G[JB_G_SIZE], half = cyl ? G[JB_G_SIZE + 1] : 0.0;
        const double ax[3] = {G[JB_G_ROT + 2], G[JB_G_ROT + 5], G[JB_G_ROT + 8]};
        const int K = cyl ? 32 : 0;
        for (int k = 0; k <= K; k++) {
            const double t = K ? (2.0 * k / K - 1.0) * half : 0.0;
            double p[3];
            for (int i = 0; i < 3; i++) p[i] = G[JB_G_CENTER + i] + t * ax[i];
            {   // thread: distance to its axis segment (the thread sits on the motor axis: the same for every motor angle)
                double d[3] = {p[0] - Gt[JB_G_CENTER], p[1] - Gt[JB_G_CENTER + 1], p[2] - Gt[JB_G_CENTER + 2]};
                double s = mc::dot(d, t_ax);
                s = s > Gt[JB_G_SIZE + 1] ? Gt[JB_G_SIZE + 1] : (s < -Gt[JB_G_SIZE + 1] ? -Gt[JB_G_SIZE + 1] : s);
                for (int i = 0; i < 3; i++) d[i] -= s * t_ax[i];
                if (sqrt(mc::dot(d, d)) < Gt[JB_G_SIZE] + rad + margin) return false;
            }
            // ellipsoid: rotate the POINT by -phi about the motor axis instead of the ellipsoid by +phi
            double r[3] = {p[0] - am[0], p[1] - am[1], p[2] - am[2]};
            const double z = mc::dot(r, em);
            double u[3] = {r[0] - z * em[0], r[1] - z * em[1], r[2] - z * em[2]}, w[3];
            mc::cross(em, u, w);
            const double infl = rad + margin + 2.5e-4;
            for (int a = 0; a < 96; a++) {
                const double phi = -2.0 * 3.141592653589793 * a / 96, cs = cos(phi), sn = sin(phi);
                double q[3], f = 0;
                for (int i = 0; i < 3; i++) q[i] = am[i] + z * em[i] + cs * u[i] + sn * w[i] - Ge[JB_G_CENTER + i];
                for (int i = 0; i < 3; i++) {
                    const double li = (Ge[JB_G_ROT + i] * q[0] + Ge[JB_G_ROT + 3 + i] * q[1] + Ge[JB_G_ROT + 6 + i] * q[2]) / (Ge[JB_G_SIZE + i] + infl);
                    f += li * li;
                }
                if (f < 1.0) return false;
            }
        }
    }
    return true;
}

}  // namespace jb

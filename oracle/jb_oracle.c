/* jb_oracle.c — CPU fp64 RESTATEMENT of the Jitterbug hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under jitterbug_amd/ may import, link or
 * execute this file; it is the checker for the HIP path (tests/, smoke(),
 * bench.py's cpu_baseline leg).
 *
 * PARITY PIN STATUS: the physics part (substep) is "parity unpinned" against
 * the reference's MuJoCo 2.0: the reference ships no golden vectors or tests
 * and MuJoCo/dm_control are absent from /root/reference and from this image
 * (SURVEY.md §8c).  It is pinned only by the closed-form known-answer tests of
 * SURVEY.md §8(c) (tests/test_oracle_kat.py) and by internal cross-checks
 * (independent numpy dynamics, conservation laws, KKT of the contact solve).
 * The observation / reward / reset parts follow the reference's own Python
 * line by line and are pinned by the KATs derived from that text.
 *
 * What is restated, and from where:
 *  - model tables: include/jitterbug_model.h   (reference: jitterbug.xml:1-148)
 *  - substep     : MuJoCo 2.0 forward dynamics + Euler step as driven by
 *                  dm_control Physics.step() (mj_step2 then mj_step1), reached
 *                  from reference jitterbug.py:84-90 (control.Environment, 50
 *                  substeps per control step).  Third-party; pipeline restated
 *                  from MuJoCo's published documentation / source:
 *                  kinematics -> joint-space inertia M -> bias (Coriolis,
 *                  centrifugal, gravity) -> passive spring/damper -> actuator
 *                  -> plane collisions -> pyramidal soft contacts (solref /
 *                  solimp impedance, diagApprox regulariser) -> convex contact
 *                  problem (MuJoCo: Newton; here: its dual by PGS run to
 *                  convergence, same unique optimum) -> semi-implicit Euler with
 *                  implicit joint damping.
 *  - accessors   : reference jitterbug.py:180-317
 *  - reset       : reference jitterbug.py:601-666 (RNG replaced by Philox4x32-10
 *                  counter streams keyed (seed, env, episode))
 *  - observation : reference jitterbug.py:673-763, tables :324-372
 *  - reward      : reference jitterbug.py:840-925 + dm_control rewards.tolerance
 *
 * Algorithmic choice: the dynamics are assembled by the projection (Kane)
 * form  M = sum_b m Jv^T Jv + Jw^T I Jw,  bias_k = sum_b Jv_k.F_b + Jw_k.N_b
 * in world axes — deliberately NOT the root-frame composite-body / Schur
 * scheme the HIP kernel uses, so that agreement between the two is evidence.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/jitterbug_model.h"

#define NB JB_NBODY
#define NV JB_NV
#define NQ JB_NQ
#define NH JB_NHINGE
#define MAXCON 64
#define MAXROW (4 * MAXCON)
#define WARM_PAIR (JB_NGEOM * 16 + JB_NV)    /* then 4 x 4: pyramid forces of the mass - upper-leg contact of leg l */
#define WARM_PAIR2 (WARM_PAIR + 16)          /* then 4 x 4: pyramid forces of the thread - upper-leg contact of leg l */
#define WARM_SIZE (JB_NGEOM * 16 + JB_NV + 32)   /* pyramid forces by (geom,slot,edge) + last qacc + the geom-geom contacts */

static const int PARENT[NB] = {-1, 0, 1, 0, 3, 0, 5, 0, 7, 0};

typedef struct {
    int contacts;        /* 1: plane contacts on                                   */
    int implicit_damp;   /* 1: Euler treats joint damping implicitly (MuJoCo)      */
    int solver_iters;    /* max PGS sweeps                                          */
    double solver_tol;   /* stop when max |df| * scale < tol (0: run all sweeps)   */
    int warmstart;       /* 1: start PGS from the forces in the warm buffer        */
    int feet_only;       /* 1: only the 4 foot spheres collide                     */
    int solver;          /* 0: dual PGS, 1: primal Newton with exact line search   */
                         /*    (MuJoCo 2.0's default solver; same unique optimum)  */
    int pair_contacts;   /* bit 0 (1): the geom-geom pairs that can touch on randomised models - the eccentric-mass ellipsoid against the
                            upper-leg cylinder of each leg - collide like MuJoCo's mjc_Convex would make them (0: floor only) */
} jbo_opts;

typedef struct {
    int ncon_last, ncon_max;
    int sweeps_total, sweeps_max;
    int nsolve;
    int overflow;
    double resid_max;
    double margin_min;   /* smallest |distance| of any contact CANDIDATE point (touching or not) seen by collide(): how close the
                            step came to a contact switching on or off exactly at a substep boundary.  Contact activation is the
                            one discontinuity of the model (dist < 0, MuJoCo margin 0): a step whose margin_min is below the
                            position error of an fp32 run can legitimately differ from it by one substep's contact impulse. */
} jbo_stats;

/* ------------------------------------------------------------------ small math */
static inline void cross3(double* o, const double* a, const double* b) {
    double x = a[1] * b[2] - a[2] * b[1], y = a[2] * b[0] - a[0] * b[2], z = a[0] * b[1] - a[1] * b[0];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline double dot3(const double* a, const double* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
static inline void matvec3(double* o, const double* R, const double* v) {
    double x = R[0] * v[0] + R[1] * v[1] + R[2] * v[2];
    double y = R[3] * v[0] + R[4] * v[1] + R[5] * v[2];
    double z = R[6] * v[0] + R[7] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void matTvec3(double* o, const double* R, const double* v) {
    double x = R[0] * v[0] + R[3] * v[1] + R[6] * v[2];
    double y = R[1] * v[0] + R[4] * v[1] + R[7] * v[2];
    double z = R[2] * v[0] + R[5] * v[1] + R[8] * v[2];
    o[0] = x; o[1] = y; o[2] = z;
}
static inline void matmul3(double* o, const double* A, const double* B) {
    double t[9];
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) t[3 * i + j] = A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j] + A[3 * i + 2] * B[6 + j];
    memcpy(o, t, sizeof t);
}
/* mju_quat2Mat (used by reference jitterbug.py:200,270,287,300) */
static void quat2mat(double* R, const double* q) {
    double w = q[0], x = q[1], y = q[2], z = q[3];
    R[0] = w * w + x * x - y * y - z * z; R[1] = 2 * (x * y - w * z);           R[2] = 2 * (x * z + w * y);
    R[3] = 2 * (x * y + w * z);           R[4] = w * w - x * x + y * y - z * z; R[5] = 2 * (y * z - w * x);
    R[6] = 2 * (x * z - w * y);           R[7] = 2 * (y * z + w * x);           R[8] = w * w - x * x - y * y + z * z;
}
static void rodrigues(double* R, const double* e, double th) {
    double c = cos(th), s = sin(th), v = 1 - c;
    R[0] = c + e[0] * e[0] * v;        R[1] = e[0] * e[1] * v - e[2] * s; R[2] = e[0] * e[2] * v + e[1] * s;
    R[3] = e[1] * e[0] * v + e[2] * s; R[4] = c + e[1] * e[1] * v;        R[5] = e[1] * e[2] * v - e[0] * s;
    R[6] = e[2] * e[0] * v - e[1] * s; R[7] = e[2] * e[1] * v + e[0] * s; R[8] = c + e[2] * e[2] * v;
}

/* ------------------------------------------------------------------ kinematics */
typedef struct {
    double R[NB][9], t[NB][3];      /* reference (root@qpos0) coords -> world: X = R x0 + t */
    double com[NB][3], Iw[NB][9];   /* world COM and inertia tensor about it, world axes */
    double anchor[NB][3], axis[NB][3]; /* world hinge anchor/axis of body b's own hinge (b>=1) */
    double mass[NB];
} Kin;

static void kinematics(const double* P, double* qpos, Kin* k) {
    /* MuJoCo normalises the free-joint quaternion in place (mj_kinematics). */
    double* q = qpos + 3;
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    if (n < 1e-15) { q[0] = 1; q[1] = q[2] = q[3] = 0; } else { for (int i = 0; i < 4; i++) q[i] /= n; }
    quat2mat(k->R[0], q);
    for (int i = 0; i < 3; i++) k->t[0][i] = qpos[i];
    for (int b = 1; b < NB; b++) {
        const double* H = P + JB_P_HINGE + (b - 1) * JB_HINGE_STRIDE;
        int p = PARENT[b];
        double Rj[9], tmp[3], a_rot[3];
        rodrigues(Rj, H + JB_H_AXIS, qpos[7 + (b - 1)]);
        matmul3(k->R[b], k->R[p], Rj);
        matvec3(a_rot, Rj, H + JB_H_ANCHOR);
        for (int i = 0; i < 3; i++) tmp[i] = H[JB_H_ANCHOR + i] - a_rot[i];
        matvec3(k->t[b], k->R[p], tmp);
        for (int i = 0; i < 3; i++) k->t[b][i] += k->t[p][i];
        matvec3(k->anchor[b], k->R[p], H + JB_H_ANCHOR);
        for (int i = 0; i < 3; i++) k->anchor[b][i] += k->t[p][i];
        matvec3(k->axis[b], k->R[p], H + JB_H_AXIS);
    }
    for (int b = 0; b < NB; b++) {
        const double* B = P + JB_P_BODY + b * JB_BODY_STRIDE;
        k->mass[b] = B[JB_B_MASS];
        matvec3(k->com[b], k->R[b], B + JB_B_COM);
        for (int i = 0; i < 3; i++) k->com[b][i] += k->t[b][i];
        const double* ii = B + JB_B_INERTIA;
        double I0[9] = {ii[0], ii[3], ii[4], ii[3], ii[1], ii[5], ii[4], ii[5], ii[2]}, T[9], Rt[9];
        matmul3(T, k->R[b], I0);
        for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) Rt[3 * i + j] = k->R[b][3 * j + i];
        matmul3(k->Iw[b], T, Rt);
    }
}

/* Jacobian columns of body b for world point x: Jv (3xNV), Jw (3xNV), column-major by dof */
static void jacobian(const Kin* k, int b, const double* x, double Jv[NV][3], double Jw[NV][3]) {
    memset(Jv, 0, sizeof(double) * NV * 3);
    memset(Jw, 0, sizeof(double) * NV * 3);
    for (int j = 0; j < 3; j++) Jv[j][j] = 1.0;            /* root translation: world axes */
    double r[3];
    for (int j = 0; j < 3; j++) {                          /* root rotation: body-local axes through root origin */
        double e[3] = {k->R[0][j], k->R[0][3 + j], k->R[0][6 + j]};
        for (int i = 0; i < 3; i++) { Jw[3 + j][i] = e[i]; r[i] = x[i] - k->t[0][i]; }
        cross3(Jv[3 + j], e, r);
    }
    for (int a = b; a > 0; a = PARENT[a]) {
        int d = 5 + a;
        for (int i = 0; i < 3; i++) { Jw[d][i] = k->axis[a][i]; r[i] = x[i] - k->anchor[a][i]; }
        cross3(Jv[d], k->axis[a], r);
    }
}

/* M (NVxNV row-major), bias (Coriolis+centrifugal+gravity) in MuJoCo dof coordinates */
static void dynamics(const double* P, const Kin* k, const double* qvel, double* M, double* bias) {
    static const double zero3[3] = {0, 0, 0};
    double Jv[NB][NV][3], Jw[NB][NV][3];
    memset(M, 0, sizeof(double) * NV * NV);
    for (int b = 0; b < NB; b++) {
        jacobian(k, b, k->com[b], Jv[b], Jw[b]);
        for (int i = 0; i < NV; i++) {
            double IJw[3];
            matvec3(IJw, k->Iw[b], Jw[b][i]);
            for (int j = 0; j <= i; j++)
                M[i * NV + j] += k->mass[b] * dot3(Jv[b][i], Jv[b][j]) + dot3(IJw, Jw[b][j]);
        }
    }
    for (int i = 0; i < NV; i++) for (int j = 0; j < i; j++) M[j * NV + i] = M[i * NV + j];

    /* velocity-product accelerations with qacc = 0, gravity folded in as -g at the root */
    const double* g = P + JB_P_GRAVITY;
    double w[NB][3], al[NB][3], ref[NB][3], Aref[NB][3];
    matvec3(w[0], k->R[0], qvel + 3);
    memcpy(al[0], zero3, sizeof zero3);
    for (int i = 0; i < 3; i++) { ref[0][i] = k->t[0][i]; Aref[0][i] = -g[i]; }
    memset(bias, 0, sizeof(double) * NV);
    for (int b = 0; b < NB; b++) {
        if (b > 0) {
            int p = PARENT[b];
            double qd = qvel[5 + b], r[3], t1[3], t2[3];
            for (int i = 0; i < 3; i++) w[b][i] = w[p][i] + k->axis[b][i] * qd;
            cross3(t1, w[p], k->axis[b]);
            for (int i = 0; i < 3; i++) al[b][i] = al[p][i] + t1[i] * qd;
            for (int i = 0; i < 3; i++) { r[i] = k->anchor[b][i] - ref[p][i]; ref[b][i] = k->anchor[b][i]; }
            cross3(t1, al[p], r);
            cross3(t2, w[p], r); cross3(t2, w[p], t2);
            for (int i = 0; i < 3; i++) Aref[b][i] = Aref[p][i] + t1[i] + t2[i];
        }
        double r[3], t1[3], t2[3], ac[3], F[3], N[3], Iw_[3], Ial[3];
        for (int i = 0; i < 3; i++) r[i] = k->com[b][i] - ref[b][i];
        cross3(t1, al[b], r);
        cross3(t2, w[b], r); cross3(t2, w[b], t2);
        for (int i = 0; i < 3; i++) { ac[i] = Aref[b][i] + t1[i] + t2[i]; F[i] = k->mass[b] * ac[i]; }
        matvec3(Iw_, k->Iw[b], w[b]);
        matvec3(Ial, k->Iw[b], al[b]);
        cross3(t1, w[b], Iw_);
        for (int i = 0; i < 3; i++) N[i] = Ial[i] + t1[i];
        for (int d = 0; d < NV; d++) bias[d] += dot3(Jv[b][d], F) + dot3(Jw[b][d], N);
    }
}

/* dense Cholesky A = L L^T (lower, in place), n<=NV; returns 0 on success */
static int chol(double* A, int n, int ld) {
    for (int j = 0; j < n; j++) {
        double s = A[j * ld + j];
        for (int k = 0; k < j; k++) s -= A[j * ld + k] * A[j * ld + k];
        if (!(s > 0)) return -1;
        double d = sqrt(s);
        A[j * ld + j] = d;
        for (int i = j + 1; i < n; i++) {
            double t = A[i * ld + j];
            for (int k = 0; k < j; k++) t -= A[i * ld + k] * A[j * ld + k];
            A[i * ld + j] = t / d;
        }
    }
    return 0;
}
static void chol_solve(const double* L, int n, int ld, double* x) {
    for (int i = 0; i < n; i++) { double s = x[i]; for (int k = 0; k < i; k++) s -= L[i * ld + k] * x[k]; x[i] = s / L[i * ld + i]; }
    for (int i = n - 1; i >= 0; i--) { double s = x[i]; for (int k = i + 1; k < n; k++) s -= L[k * ld + i] * x[k]; x[i] = s / L[i * ld + i]; }
}

/* ------------------------------------------------------------------ collisions */
/* body: the body the contact Jacobian is ADDED for (the plane contacts' only body; geom2's body of a geom-geom contact);
 * body2: the body it is SUBTRACTED for (-1: the world); n: unit normal, world axes, pointing from geom1 to geom2 (MuJoCo's
 * convention; +z for the floor); wslot: where the contact's pyramid forces are kept in the warm-start buffer. */
typedef struct { double dist, pos[3], n[3]; int body, body2, geom, slot, wslot; } Contact;

/* ---- geom-geom narrow phase: Minkowski Portal Refinement ------------------------------------------------------------------
 * MuJoCo 2.0 sends every pair without a dedicated routine (ellipsoid-cylinder among them) through mjc_Convex, i.e. libccd's
 * ccdMPRPenetration (G. Snethen's XenoCollide algorithm) with opt.mpr_tolerance = 1e-6 and opt.mpr_iterations = 50, over
 * MuJoCo's support functions.  Neither library is under /root/reference; what follows restates the published algorithm
 * (libccd src/mpr.c: discoverPortal / refinePortal / findPenetr / findPos / expandPortal, same tests in the same order)
 * and MuJoCo's ellipsoid and cylinder support mappings.  Result convention (libccd): translating obj2 by depth * dir separates
 * the two, i.e. dir points from obj1 to obj2 - MuJoCo copies it into the contact normal - and pos is the mean of the two
 * witness points interpolated over the final portal. */
typedef struct { int type; double c[3], R[9], sz[3]; } CGeom;
typedef struct { double v[3], v1[3], v2[3]; } MSupp;
#define MPR_EPS 2.220446049250313e-16
#define MPR_TOL 1e-6
#define MPR_ITERS 50
static void cgeom_support(const CGeom* g, const double* d, double* out) {
    double dl[3], sl[3];
    matTvec3(dl, g->R, d);
    if (g->type == JB_GEOM_ELLIPSOID) {
        double den = sqrt(g->sz[0] * g->sz[0] * dl[0] * dl[0] + g->sz[1] * g->sz[1] * dl[1] * dl[1] + g->sz[2] * g->sz[2] * dl[2] * dl[2]);
        for (int i = 0; i < 3; i++) sl[i] = den > 1e-15 ? g->sz[i] * g->sz[i] * dl[i] / den : 0.0;
    } else {        /* cylinder: size = radius, half length; axis = geom z */
        double nr = sqrt(dl[0] * dl[0] + dl[1] * dl[1]);
        sl[0] = nr > 1e-15 ? dl[0] / nr * g->sz[0] : 0.0;
        sl[1] = nr > 1e-15 ? dl[1] / nr * g->sz[0] : 0.0;
        sl[2] = dl[2] > 0 ? g->sz[1] : (dl[2] < 0 ? -g->sz[1] : 0.0);
    }
    matvec3(out, g->R, sl);
    for (int i = 0; i < 3; i++) out[i] += g->c[i];
}
static void mpr_support(const CGeom* a, const CGeom* b, const double* dir, MSupp* s) {
    double nd[3] = {-dir[0], -dir[1], -dir[2]};
    cgeom_support(a, dir, s->v1);
    cgeom_support(b, nd, s->v2);
    for (int i = 0; i < 3; i++) s->v[i] = s->v1[i] - s->v2[i];
}
static int mpr_zero(double x) { return fabs(x) < MPR_EPS; }
static int mpr_eq(double a, double b) {
    double ab = fabs(a - b);
    if (ab < MPR_EPS) return 1;
    double aa = fabs(a), bb = fabs(b);
    return ab < MPR_EPS * (bb > aa ? bb : aa);
}
static void vnormalize(double* v) { double n = sqrt(dot3(v, v)); for (int i = 0; i < 3; i++) v[i] /= n; }
static void portal_dir(const MSupp* ps, double* dir) {
    double a[3], b[3];
    for (int i = 0; i < 3; i++) { a[i] = ps[2].v[i] - ps[1].v[i]; b[i] = ps[3].v[i] - ps[1].v[i]; }
    cross3(dir, a, b);
    vnormalize(dir);
}
static double g_mpr_tol = MPR_TOL;     /* (tests vary these to show what MuJoCo's setting leaves unconverged) */
static int g_mpr_iters = MPR_ITERS;
static int portal_reach_tolerance(const MSupp* ps, const MSupp* v4, const double* dir) {
    double dv1 = dot3(ps[1].v, dir), dv2 = dot3(ps[2].v, dir), dv3 = dot3(ps[3].v, dir), dv4 = dot3(v4->v, dir);
    double d1 = dv4 - dv1, d2 = dv4 - dv2, d3 = dv4 - dv3;
    double m = d1 < d2 ? d1 : d2;
    m = m < d3 ? m : d3;
    return mpr_eq(m, g_mpr_tol) || m < g_mpr_tol;
}
static void expand_portal(MSupp* ps, const MSupp* v4) {
    double v4v0[3];
    cross3(v4v0, v4->v, ps[0].v);
    if (dot3(ps[1].v, v4v0) > 0) {
        if (dot3(ps[2].v, v4v0) > 0) ps[1] = *v4; else ps[3] = *v4;
    } else {
        if (dot3(ps[3].v, v4v0) > 0) ps[2] = *v4; else ps[1] = *v4;
    }
}
/* closest point of triangle (a, b, c) to the origin (Ericson, Real-Time Collision Detection 5.1.5): what libccd's
 * ccdVec3PointTriDist2(origin, ...) returns as witness */
static void tri_closest_to_origin(const double* a, const double* b, const double* c, double* out) {
    double ab[3], ac[3], ap[3], bp[3], cp[3];
    for (int i = 0; i < 3; i++) { ab[i] = b[i] - a[i]; ac[i] = c[i] - a[i]; ap[i] = -a[i]; bp[i] = -b[i]; cp[i] = -c[i]; }
    double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    if (d1 <= 0 && d2 <= 0) { memcpy(out, a, 24); return; }
    double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
    if (d3 >= 0 && d4 <= d3) { memcpy(out, b, 24); return; }
    double vc = d1 * d4 - d3 * d2;
    if (vc <= 0 && d1 >= 0 && d3 <= 0) { double v = d1 / (d1 - d3); for (int i = 0; i < 3; i++) out[i] = a[i] + v * ab[i]; return; }
    double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
    if (d6 >= 0 && d5 <= d6) { memcpy(out, c, 24); return; }
    double vb = d5 * d2 - d1 * d6;
    if (vb <= 0 && d2 >= 0 && d6 <= 0) { double w = d2 / (d2 - d6); for (int i = 0; i < 3; i++) out[i] = a[i] + w * ac[i]; return; }
    double va = d3 * d6 - d5 * d4;
    if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
        double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
        for (int i = 0; i < 3; i++) out[i] = b[i] + w * (c[i] - b[i]);
        return;
    }
    double den = 1.0 / (va + vb + vc), v = vb * den, w = vc * den;
    for (int i = 0; i < 3; i++) out[i] = a[i] + ab[i] * v + ac[i] * w;
}
/* returns 0 with depth / dir / pos when the two geoms intersect, -1 otherwise */
static int mpr_penetration(const CGeom* o1, const CGeom* o2, double* depth, double* dir_out, double* pos) {
    MSupp ps[4], v4;
    double dir[3], va[3], vb[3], dot;
    /* discoverPortal */
    memcpy(ps[0].v1, o1->c, 24); memcpy(ps[0].v2, o2->c, 24);
    for (int i = 0; i < 3; i++) ps[0].v[i] = ps[0].v1[i] - ps[0].v2[i];
    if (mpr_eq(ps[0].v[0], 0) && mpr_eq(ps[0].v[1], 0) && mpr_eq(ps[0].v[2], 0)) ps[0].v[0] += MPR_EPS * 10;
    for (int i = 0; i < 3; i++) dir[i] = -ps[0].v[i];
    vnormalize(dir);
    mpr_support(o1, o2, dir, &ps[1]);
    dot = dot3(ps[1].v, dir);
    if (mpr_zero(dot) || dot < 0) return -1;
    cross3(dir, ps[0].v, ps[1].v);
    if (mpr_zero(dot3(dir, dir))) {
        /* the origin lies on the segment v0-v1 (or on v1): libccd's findPenetrSegment / findPenetrTouch */
        *depth = sqrt(dot3(ps[1].v, ps[1].v));
        memcpy(dir_out, ps[1].v, 24);
        if (*depth > 0) vnormalize(dir_out);
        for (int i = 0; i < 3; i++) pos[i] = 0.5 * (ps[1].v1[i] + ps[1].v2[i]);
        return 0;
    }
    vnormalize(dir);
    mpr_support(o1, o2, dir, &ps[2]);
    dot = dot3(ps[2].v, dir);
    if (mpr_zero(dot) || dot < 0) return -1;
    for (int i = 0; i < 3; i++) { va[i] = ps[1].v[i] - ps[0].v[i]; vb[i] = ps[2].v[i] - ps[0].v[i]; }
    cross3(dir, va, vb);
    vnormalize(dir);
    if (dot3(dir, ps[0].v) > 0) { MSupp t = ps[1]; ps[1] = ps[2]; ps[2] = t; for (int i = 0; i < 3; i++) dir[i] = -dir[i]; }
    for (int guard = 0;; guard++) {
        if (guard > 100) return -1;
        mpr_support(o1, o2, dir, &ps[3]);
        dot = dot3(ps[3].v, dir);
        if (mpr_zero(dot) || dot < 0) return -1;
        int cont = 0;
        cross3(va, ps[1].v, ps[3].v);
        dot = dot3(va, ps[0].v);
        if (dot < 0 && !mpr_zero(dot)) { ps[2] = ps[3]; cont = 1; }
        if (!cont) {
            cross3(va, ps[3].v, ps[2].v);
            dot = dot3(va, ps[0].v);
            if (dot < 0 && !mpr_zero(dot)) { ps[1] = ps[3]; cont = 1; }
        }
        if (!cont) break;
        for (int i = 0; i < 3; i++) { va[i] = ps[1].v[i] - ps[0].v[i]; vb[i] = ps[2].v[i] - ps[0].v[i]; }
        cross3(dir, va, vb);
        vnormalize(dir);
    }
    /* refinePortal */
    for (int guard = 0;; guard++) {
        if (guard > 1000) return -1;
        portal_dir(ps, dir);
        dot = dot3(dir, ps[1].v);
        if (mpr_zero(dot) || dot > 0) break;                      /* the portal encapsules the origin */
        mpr_support(o1, o2, dir, &v4);
        dot = dot3(v4.v, dir);
        if (!(mpr_zero(dot) || dot > 0) || portal_reach_tolerance(ps, &v4, dir)) return -1;
        expand_portal(ps, &v4);
    }
    /* findPenetr */
    for (int it = 0;; it++) {
        portal_dir(ps, dir);
        mpr_support(o1, o2, dir, &v4);
        if (portal_reach_tolerance(ps, &v4, dir) || it > g_mpr_iters) {
            double pd[3];
            tri_closest_to_origin(ps[1].v, ps[2].v, ps[3].v, pd);
            *depth = sqrt(dot3(pd, pd));
            if (mpr_zero(pd[0]) && mpr_zero(pd[1]) && mpr_zero(pd[2])) memcpy(pd, dir, 24);
            vnormalize(pd);
            memcpy(dir_out, pd, 24);
            /* findPos: barycentric coordinates of the origin in the tetrahedron (v0, v1, v2, v3) */
            double b[4], vec[3], sum;
            portal_dir(ps, dir);
            cross3(vec, ps[1].v, ps[2].v); b[0] = dot3(vec, ps[3].v);
            cross3(vec, ps[3].v, ps[2].v); b[1] = dot3(vec, ps[0].v);
            cross3(vec, ps[0].v, ps[1].v); b[2] = dot3(vec, ps[3].v);
            cross3(vec, ps[2].v, ps[1].v); b[3] = dot3(vec, ps[0].v);
            sum = b[0] + b[1] + b[2] + b[3];
            if (mpr_zero(sum) || sum < 0) {
                b[0] = 0;
                cross3(vec, ps[2].v, ps[3].v); b[1] = dot3(vec, dir);
                cross3(vec, ps[3].v, ps[1].v); b[2] = dot3(vec, dir);
                cross3(vec, ps[1].v, ps[2].v); b[3] = dot3(vec, dir);
                sum = b[1] + b[2] + b[3];
            }
            double p1[3] = {0, 0, 0}, p2[3] = {0, 0, 0};
            for (int k2 = 0; k2 < 4; k2++) for (int i = 0; i < 3; i++) { p1[i] += b[k2] * ps[k2].v1[i]; p2[i] += b[k2] * ps[k2].v2[i]; }
            for (int i = 0; i < 3; i++) pos[i] = 0.5 * (p1[i] + p2[i]) / sum;
            return 0;
        }
        expand_portal(ps, &v4);
    }
}

/* ---- the narrow phase the oracle (and the HIP kernel) USE for that pair: the geometric contact --------------------------------
 * MuJoCo's MPR stops at a portal 1e-6 wide; on this pair (an ellipsoid of ~5 mm against a cylinder of 1.5 mm radius, overlaps of
 * 0.02-0.2 mm) that leaves the normal undetermined to +-10 degrees and the depth to a few per cent from one configuration to the
 * next, and the iteration converges too slowly to be run to a reproducible limit (tolerance 1e-10: normal still 4e-3 off,
 * tests/test_pair_contact.py) - no independent implementation, let alone an fp32 one, can reproduce its output.  What its output
 * scatters around is the geometric contact of the two bodies, which is what is computed here in closed form + two nested
 * one-dimensional Newton / secant iterations, smooth in the configuration:
 *     x* = the point of the cylinder's AXIS SEGMENT with the smallest SIGNED distance sd to the ellipsoid (negative inside it), q* the
 *     ellipsoid's nearest point to x*,   n = the ellipsoid's outward normal at q*  (mass -> leg),
 *     dist = sd - r_cyl  (negative: penetration),   pos = midpoint of q* and the cylinder surface point x* - r n.
 * (An overlap at the very end of the leg is taken against the segment's end point - a rounded end where the cylinder has a
 * flat cap; the upper legs' ends lie inside the shoulder / knee geometry.) */
/* nearest point q of the axis-aligned ellipsoid (semi-axes s) to a point y, outside OR inside it: q_i = s_i^2 y_i / (s_i^2 + lam)
 * with lam the root of F(lam) = sum_i (s_i y_i / (s_i^2 + lam))^2 - 1 in (-min s_i^2, inf): positive outside, negative inside.
 * F is convex and decreasing there: Newton, kept inside the bracket it has established.  Returns the SIGNED distance
 * (negative inside) and the outward unit normal at q. */
static double ellipsoid_nearest(const double* s, const double* y, double* q, double* nrm) {
    double smin2 = s[0] * s[0];
    for (int i = 1; i < 3; i++) if (s[i] * s[i] < smin2) smin2 = s[i] * s[i];
    double F0 = -1;
    for (int i = 0; i < 3; i++) F0 += (y[i] / s[i]) * (y[i] / s[i]);
    double lo, hi, lam = 0;                      /* F(lo) > 0 > F(hi) */
    if (F0 > 0) { lo = 0; hi = INFINITY; } else { lo = -smin2; hi = 0; }
    for (int it = 0; it < 100; it++) {
        double F = -1, dF = 0;
        for (int i = 0; i < 3; i++) { double a = s[i] * s[i] + lam, w = s[i] * y[i] / a; F += w * w; dF -= 2 * w * w / a; }
        if (F > 0) lo = lam; else hi = lam;
        double nl = (dF < 0) ? lam - F / dF : lam;
        if (!(nl > lo && nl < hi)) nl = isinf(hi) ? 2 * lam + smin2 : 0.5 * (lo + hi);
        if (fabs(nl - lam) <= 1e-16 * (smin2 + fabs(lam))) { lam = nl; break; }
        lam = nl;
    }
    double g[3], gl = 0, d2 = 0;
    for (int i = 0; i < 3; i++) { q[i] = s[i] * s[i] * y[i] / (s[i] * s[i] + lam); g[i] = q[i] / (s[i] * s[i]); gl += g[i] * g[i]; d2 += (y[i] - q[i]) * (y[i] - q[i]); }
    gl = sqrt(gl);
    for (int i = 0; i < 3; i++) nrm[i] = g[i] / gl;                  /* gradient of sum (x_i / s_i)^2 at q */
    return lam >= 0 ? sqrt(d2) : -sqrt(d2);
}
/* returns 1 with dist (<0: penetration), n (ellipsoid -> cylinder), pos, all in world axes */
static int pair_geometric_exact(const CGeom* e, const CGeom* c, double* dist, double* n, double* pos) {
    double cl[3], ul[3], d[3], ax[3] = {c->R[2], c->R[5], c->R[8]};
    for (int i = 0; i < 3; i++) d[i] = c->c[i] - e->c[i];
    matTvec3(cl, e->R, d);
    matTvec3(ul, e->R, ax);
    const double h = c->sz[1];
    /* f(t) = u . n_out(x(t)) is the derivative of the (convex) signed distance along the axis: monotone, root by safeguarded secant */
    double ta = -h, tb = h, fa, fb, x[3], q[3], nl[3];
    for (int i = 0; i < 3; i++) x[i] = cl[i] + ta * ul[i];
    ellipsoid_nearest(e->sz, x, q, nl);
    fa = dot3(nl, ul);
    for (int i = 0; i < 3; i++) x[i] = cl[i] + tb * ul[i];
    ellipsoid_nearest(e->sz, x, q, nl);
    fb = dot3(nl, ul);
    double t;
    if (fa >= 0) t = ta; else if (fb <= 0) t = tb;
    else {
        t = ta - fa * (tb - ta) / (fb - fa);
        for (int it = 0; it < 200; it++) {
            for (int i = 0; i < 3; i++) x[i] = cl[i] + t * ul[i];
            ellipsoid_nearest(e->sz, x, q, nl);
            double f = dot3(nl, ul);
            if (f < 0) { ta = t; fa = f; } else { tb = t; fb = f; }
            if (fabs(f) < 1e-15 || tb - ta < 1e-15) break;
            double tn = t - f * (tb - ta) / (fb - fa);              /* secant through the bracket ends ... */
            if (!(tn > ta && tn < tb) || it % 3 == 2) tn = 0.5 * (ta + tb);     /* ... bisection when it leaves the bracket or stalls on one side */
            t = tn;
        }
    }
    for (int i = 0; i < 3; i++) x[i] = cl[i] + t * ul[i];
    double sd = ellipsoid_nearest(e->sz, x, q, nl), pl[3];
    *dist = sd - c->sz[0];
    for (int i = 0; i < 3; i++) pl[i] = 0.5 * (q[i] + x[i] - c->sz[0] * nl[i]);
    matvec3(n, e->R, nl);
    matvec3(pos, e->R, pl);
    for (int i = 0; i < 3; i++) pos[i] += e->c[i];
    return 1;
}
/* The same contact by the FIXED scheme the HIP kernel runs (jb_sim.hpp pair_narrow: all lanes of a wave walk through it together, so its
 * iteration counts are constants): multiplier by Newton on the secular form 1 / N(lam) - 1, clamped at -0.95 min s^2; axis parameter
 * from the bracket [t0 - s_max, t0 + s_max] (t0: the ellipsoid centre's projection) by 5 Illinois + 3 Newton steps.  On contacts
 * shallower than the cylinder radius it agrees with pair_geometric_exact to round-off (tests/test_pair_contact.py); on overlaps so deep
 * that the leg's axis runs through the mass (a robot that could not exist) both stay finite, and this is the definition. */
static void ell_lambda_fixed(const double* s2, double lam_min, const double* y, double* lam, int iters) {
    double p[3] = {s2[0] * y[0] * y[0], s2[1] * y[1] * y[1], s2[2] * y[2] * y[2]};
    for (int it = 0; it < iters; it++) {
        double N2 = 0, S3 = 0;
        for (int i = 0; i < 3; i++) { double ia = 1.0 / (s2[i] + *lam), t = p[i] * ia * ia; N2 += t; S3 += t * ia; }
        if (N2 < 1e-30) N2 = 1e-30;
        if (S3 < 1e-30) S3 = 1e-30;
        double N = sqrt(N2), nl = *lam + (N - 1.0) * N2 / S3;
        *lam = nl > lam_min ? nl : lam_min;
    }
}
typedef struct { double cl[3], ul[3], s2[3], lam_min, lam, x[3], g[3]; } PairEval;
static double pair_eval(PairEval* w, double t, int iters) {
    for (int i = 0; i < 3; i++) w->x[i] = w->cl[i] + t * w->ul[i];
    ell_lambda_fixed(w->s2, w->lam_min, w->x, &w->lam, iters);
    for (int i = 0; i < 3; i++) w->g[i] = w->x[i] / (w->s2[i] + w->lam);
    return dot3(w->g, w->ul) / sqrt(dot3(w->g, w->g));
}
static int pair_geometric(const CGeom* e, const CGeom* c, double* dist, double* n, double* pos) {
    PairEval w;
    double d[3], ax[3] = {c->R[2], c->R[5], c->R[8]};
    for (int i = 0; i < 3; i++) { d[i] = c->c[i] - e->c[i]; w.s2[i] = e->sz[i] * e->sz[i]; }
    matTvec3(w.cl, e->R, d);
    matTvec3(w.ul, e->R, ax);
    double smin2 = fmin(w.s2[0], fmin(w.s2[1], w.s2[2])), smax = fmax(e->sz[0], fmax(e->sz[1], e->sz[2]));
    const double half = c->sz[1], rad = c->sz[0];
    w.lam_min = -0.95 * smin2; w.lam = 0;
    double t0 = -dot3(w.cl, w.ul);
    double a0 = fmax(t0 - smax, -half), b0 = fmin(fmax(t0 + smax, -half), half);
    double ta = a0, tb = fmax(b0, a0);
    double fa = pair_eval(&w, ta, 8), fb = pair_eval(&w, tb, 6);
    int at_a = !(fa < 0), at_b = !(fb > 0);
    double tc = tb;
    for (int it = 0; it < 5; it++) {
        double den = fb - fa;
        int ok = fabs(den) > 1e-20;
        tc = ok ? tb - fb * (tb - ta) / den : tb;
        tc = fmin(fmax(tc, fmin(ta, tb)), fmax(ta, tb));
        double fc = pair_eval(&w, tc, 4);
        if (fc * fb < 0) { ta = tb; fa = fb; } else fa *= 0.5;
        tb = tc; fb = fc;
    }
    double lo_t = fmin(ta, tb), hi_t = fmax(ta, tb), fc = fb;
    for (int it = 0; it < 3; it++) {
        double A[3], Au[3], Ag[3], dq[3], dg[3];
        for (int i = 0; i < 3; i++) { A[i] = w.s2[i] / (w.s2[i] + w.lam); Au[i] = A[i] * w.ul[i]; Ag[i] = A[i] * w.g[i]; }
        double dlam = dot3(w.g, Au) / dot3(w.g, Ag);
        for (int i = 0; i < 3; i++) { dq[i] = Au[i] - Ag[i] * dlam; dg[i] = dq[i] / w.s2[i]; }
        double gg = dot3(w.g, w.g), ig = 1.0 / sqrt(gg);
        double ndg = dot3(w.g, dg) * ig, df = (dot3(w.ul, dg) - (dot3(w.g, w.ul) * ig) * ndg) * ig;
        if (df > 1e-12) tc = tc - fc / df;
        tc = fmin(fmax(tc, lo_t), hi_t);
        fc = pair_eval(&w, tc, 4);
    }
    tc = at_a ? a0 : (at_b ? fmax(b0, a0) : tc);
    (void)pair_eval(&w, tc, 5);
    double gg = dot3(w.g, w.g), gl = sqrt(gg), nl[3], pl[3];
    *dist = w.lam * gl - rad;
    for (int i = 0; i < 3; i++) { nl[i] = w.g[i] / gl; pl[i] = w.x[i] - 0.5 * w.lam * w.g[i] - 0.5 * rad * nl[i]; }
    matvec3(n, e->R, nl);
    matvec3(pos, e->R, pl);
    for (int i = 0; i < 3; i++) pos[i] += e->c[i];
    return 1;
}
/* ---- the second geom-geom pair randomised models can bring together: the motor-axis THREAD (geom 20, a cylinder of 1 mm
 * radius on the motor body, coaxial with the motor hinge) against the upper-leg cylinders.  MuJoCo sends cylinder - cylinder through
 * mjc_Convex = MPR like the pair above; what is simulated is again the geometric contact, with the leg - a thin rod, radius 0.61 mm - taken
 * as its axis segment:
 *     x* = the point of the LEG's axis segment with the smallest SIGNED distance sd to the thread cylinder (flat caps, closed form below),
 *     q* the thread's nearest surface point,  m = the thread's outward normal there,
 *     dist = sd - r_leg,   pos = midpoint of q* and the leg surface point x* - r_leg m,   normal (geom1 = leg -> geom2 = thread) = -m.
 * The signed distance to a convex body is convex along a line, so f(t) = m(x(t)) . u is monotone: bisection on its sign. */
/* signed distance of y (cylinder's own axes, z = its axis) to a finite cylinder (radius R, half length H); q: nearest surface point, nrm: outward normal */
static double cylinder_nearest(double R, double H, const double* y, double* q, double* nrm) {
    double rho = sqrt(y[0] * y[0] + y[1] * y[1]), er[2] = {1, 0};
    if (rho > 1e-300) { er[0] = y[0] / rho; er[1] = y[1] / rho; }
    double dr = rho - R, az = fabs(y[2]), dz = az - H, sz = y[2] < 0 ? -1.0 : 1.0;
    if (dr > 0 && dz > 0) {          /* beyond the rim */
        double d = sqrt(dr * dr + dz * dz);
        q[0] = R * er[0]; q[1] = R * er[1]; q[2] = sz * H;
        nrm[0] = dr * er[0] / d; nrm[1] = dr * er[1] / d; nrm[2] = sz * dz / d;
        return d;
    }
    if (dr > dz) { q[0] = R * er[0]; q[1] = R * er[1]; q[2] = y[2]; nrm[0] = er[0]; nrm[1] = er[1]; nrm[2] = 0; return dr; }      /* the side is nearest (outside or inside) */
    q[0] = y[0]; q[1] = y[1]; q[2] = sz * H; nrm[0] = 0; nrm[1] = 0; nrm[2] = sz;
    return dz;                                                                                                                      /* a cap is nearest */
}
#ifndef JB_THREAD_BISECT
#define JB_THREAD_BISECT 26      /* bisection steps on the leg's axis parameter: the HIP kernel (jb_sim.hpp thread_narrow) runs the same fixed count */
#endif
/* th: the thread cylinder, c: the leg cylinder (world poses); returns dist, n (leg -> thread), pos in world axes */
static void pair_thread_geometric(const CGeom* th, const CGeom* c, int steps, double* dist, double* n, double* pos) {
    double cl[3], ul[3], d[3], ax[3] = {c->R[2], c->R[5], c->R[8]};
    for (int i = 0; i < 3; i++) d[i] = c->c[i] - th->c[i];
    matTvec3(cl, th->R, d);
    matTvec3(ul, th->R, ax);
    const double h = c->sz[1], R = th->sz[0], H = th->sz[1];
    double x[3], q[3], m[3], ta = -h, tb = h, t;
    for (int i = 0; i < 3; i++) x[i] = cl[i] + ta * ul[i];
    cylinder_nearest(R, H, x, q, m);
    double fa = dot3(m, ul);
    for (int i = 0; i < 3; i++) x[i] = cl[i] + tb * ul[i];
    cylinder_nearest(R, H, x, q, m);
    double fb = dot3(m, ul);
    if (!(fa < 0)) t = ta;
    else if (!(fb > 0)) t = tb;
    else {
        for (int it = 0; it < steps; it++) {
            t = 0.5 * (ta + tb);
            for (int i = 0; i < 3; i++) x[i] = cl[i] + t * ul[i];
            cylinder_nearest(R, H, x, q, m);
            if (dot3(m, ul) < 0) ta = t; else tb = t;
        }
        t = 0.5 * (ta + tb);
    }
    for (int i = 0; i < 3; i++) x[i] = cl[i] + t * ul[i];
    double sd = cylinder_nearest(R, H, x, q, m), pl[3], nl[3];
    *dist = sd - c->sz[0];
    for (int i = 0; i < 3; i++) { pl[i] = 0.5 * (q[i] + x[i] - c->sz[0] * m[i]); nl[i] = -m[i]; }
    matvec3(n, th->R, nl);
    matvec3(pos, th->R, pl);
    for (int i = 0; i < 3; i++) pos[i] += th->c[i];
}
/* smallest distance between two segments c1 + s u1 (|s| <= h1), c2 + t u2 (|t| <= h2), unit directions (Ericson, Real-Time Collision Detection 5.1.9) */
static double segment_distance(const double* c1, const double* u1, double h1, const double* c2, const double* u2, double h2) {
    double r[3] = {c1[0] - c2[0], c1[1] - c2[1], c1[2] - c2[2]};
    double b = dot3(u1, u2), c = dot3(u1, r), f = dot3(u2, r), den = 1.0 - b * b, s, t;
    s = den > 1e-12 ? (b * f - c) / den : 0.0;
    s = s < -h1 ? -h1 : (s > h1 ? h1 : s);
    t = b * s + f;
    if (t < -h2) { t = -h2; s = b * t - c; s = s < -h1 ? -h1 : (s > h1 ? h1 : s); }
    else if (t > h2) { t = h2; s = b * t - c; s = s < -h1 ? -h1 : (s > h1 ? h1 : s); }
    double dd[3];
    for (int i = 0; i < 3; i++) dd[i] = r[i] + s * u1[i] - t * u2[i];
    return sqrt(dot3(dd, dd));
}
static void cgeom_world(const double* P, const Kin* k, int g, CGeom* out) {
    const double* G = P + JB_P_GEOM + g * JB_GEOM_STRIDE;
    int b = (int)G[JB_G_BODY];
    out->type = (int)G[JB_G_TYPE];
    matvec3(out->c, k->R[b], G + JB_G_CENTER);
    for (int i = 0; i < 3; i++) { out->c[i] += k->t[b][i]; out->sz[i] = G[JB_G_SIZE + i]; }
    matmul3(out->R, k->R[b], G + JB_G_ROT);
}
/* MuJoCo's bounding-sphere radius of a geom (rbound) */
static double cgeom_rbound(const CGeom* g) {
    if (g->type == JB_GEOM_ELLIPSOID) { double m = g->sz[0] > g->sz[1] ? g->sz[0] : g->sz[1]; return m > g->sz[2] ? m : g->sz[2]; }
    return sqrt(g->sz[0] * g->sz[0] + g->sz[1] * g->sz[1]);
}
double jbo_posed_geom_distance(int ta, const double* ca, const double* Ra, const double* sa, int tb, const double* cb, const double* Rb, const double* sb);   /* jb_clearance.c: exact GJK distance */

/* All against the floor plane z=0 with normal +z (reference: jitterbug.xml:30).
 * Restates MuJoCo's mjc_PlaneSphere / mjc_PlaneCylinder / mjc_PlaneBox /
 * mjc_PlaneConvex(ellipsoid support point).  A contact exists when dist < 0
 * (margin 0). */
static int add_contact(Contact* c, int n, double dist, const double* pos, int body, int geom, int slot) {
    if (n >= MAXCON) return n;
    c[n].dist = dist; c[n].pos[0] = pos[0]; c[n].pos[1] = pos[1]; c[n].pos[2] = pos[2];
    c[n].body = body; c[n].geom = geom; c[n].slot = slot;
    c[n].body2 = -1; c[n].n[0] = 0; c[n].n[1] = 0; c[n].n[2] = 1; c[n].wslot = (geom * 4 + slot) * 4;
    return n + 1;
}
#define MARGIN(d) do { double _a = fabs(d); if (_a < *margin) *margin = _a; } while (0)
static int collide(const double* P, const Kin* k, int feet_only, int pair_contacts, Contact* con, int* overflow, double* margin) {
    static const double nz[3] = {0, 0, 1};
    int n = 0;
    for (int g = 0; g < JB_NGEOM; g++) {
        const double* G = P + JB_P_GEOM + g * JB_GEOM_STRIDE;
        int type = (int)G[JB_G_TYPE], b = (int)G[JB_G_BODY];
        if (feet_only && !(g >= 4 && g < 20 && ((g - 4) & 3) == 3)) continue;
        double c[3], Rg[9], pos[3];
        matvec3(c, k->R[b], G + JB_G_CENTER);
        for (int i = 0; i < 3; i++) c[i] += k->t[b][i];
        matmul3(Rg, k->R[b], G + JB_G_ROT);
        const double* sz = G + JB_G_SIZE;
        int n0 = n;
        if (type == JB_GEOM_SPHERE) {
            double dist = c[2] - sz[0];
            MARGIN(dist);
            if (dist < 0) { for (int i = 0; i < 3; i++) pos[i] = c[i] - nz[i] * (sz[0] + 0.5 * dist); n = add_contact(con, n, dist, pos, b, g, 0); }
        } else if (type == JB_GEOM_ELLIPSOID) {
            /* support point of the ellipsoid in direction -n */
            double dl[3], mnz[3] = {0, 0, -1}, s[3], sw[3];
            matTvec3(dl, Rg, mnz);
            double den = sqrt(sz[0] * sz[0] * dl[0] * dl[0] + sz[1] * sz[1] * dl[1] * dl[1] + sz[2] * sz[2] * dl[2] * dl[2]);
            for (int i = 0; i < 3; i++) s[i] = sz[i] * sz[i] * dl[i] / den;
            matvec3(sw, Rg, s);
            for (int i = 0; i < 3; i++) sw[i] += c[i];
            double dist = sw[2];
            MARGIN(dist);
            if (dist < 0) { for (int i = 0; i < 3; i++) pos[i] = sw[i] - nz[i] * 0.5 * dist; n = add_contact(con, n, dist, pos, b, g, 0); }
        } else if (type == JB_GEOM_BOX) {
            int cnt = 0;
            for (int v = 0; v < 8 && cnt < 4; v++) {
                double l[3] = {(v & 1 ? sz[0] : -sz[0]), (v & 2 ? sz[1] : -sz[1]), (v & 4 ? sz[2] : -sz[2])}, vec[3];
                matvec3(vec, Rg, l);
                double dist = c[2] + vec[2];
                MARGIN(dist);
                if (dist < 0) {
                    for (int i = 0; i < 3; i++) pos[i] = c[i] + vec[i] - nz[i] * 0.5 * dist;
                    n = add_contact(con, n, dist, pos, b, g, cnt); cnt++;
                }
            }
        } else { /* cylinder */
            double axis[3] = {Rg[2], Rg[5], Rg[8]}, vec[3];
            double prjaxis = axis[2];
            if (prjaxis > 0) { for (int i = 0; i < 3; i++) axis[i] = -axis[i]; prjaxis = -prjaxis; }
            double dist0 = c[2];
            for (int i = 0; i < 3; i++) vec[i] = axis[i] * prjaxis - nz[i];
            double len2 = dot3(vec, vec);
            if (len2 >= 1e-30) { double s = sz[0] / sqrt(len2); for (int i = 0; i < 3; i++) vec[i] *= s; }
            else { vec[0] = Rg[0] * sz[0]; vec[1] = Rg[3] * sz[0]; vec[2] = Rg[6] * sz[0]; }
            double prjvec = vec[2];
            double ax[3] = {axis[0] * sz[1], axis[1] * sz[1], axis[2] * sz[1]};
            prjaxis *= sz[1];
            double d1 = dist0 + prjaxis + prjvec;
            MARGIN(d1);
            if (d1 < 0) {
                /* a second switch of this routine: which END of the cylinder is the "lower" one (prjaxis changes sign when the cylinder
                 * passes through horizontal - a leg lying flat on the floor); the points at d3 then jump to the other end.  Its distance
                 * from switching, as a length: the height of the axis end above the centre. */
                MARGIN(prjaxis);
                for (int i = 0; i < 3; i++) pos[i] = c[i] + vec[i] + ax[i] - nz[i] * 0.5 * d1;
                n = add_contact(con, n, d1, pos, b, g, 0);
                double d2 = dist0 - prjaxis + prjvec;
                MARGIN(d2);
                if (d2 < 0) {
                    for (int i = 0; i < 3; i++) pos[i] = c[i] + vec[i] - ax[i] - nz[i] * 0.5 * d2;
                    n = add_contact(con, n, d2, pos, b, g, 1);
                }
                double prjvec1 = -0.5 * prjvec, d3 = dist0 + prjaxis + prjvec1;
                MARGIN(d3);
                if (d3 < 0) {
                    double vec1[3];
                    cross3(vec1, vec, ax);
                    double l = sqrt(dot3(vec1, vec1));
                    if (l > 1e-300) { double s = sz[0] * sqrt(3.0) * 0.5 / l; for (int i = 0; i < 3; i++) vec1[i] *= s; }
                    for (int i = 0; i < 3; i++) pos[i] = c[i] + vec1[i] + ax[i] - 0.5 * vec[i] - nz[i] * 0.5 * d3;
                    n = add_contact(con, n, d3, pos, b, g, 2);
                    for (int i = 0; i < 3; i++) pos[i] = c[i] - vec1[i] + ax[i] - 0.5 * vec[i] - nz[i] * 0.5 * d3;
                    n = add_contact(con, n, d3, pos, b, g, 3);
                }
            }
        }
        if (n == MAXCON && n0 != n) *overflow = 1;
    }
    if ((pair_contacts & 2) && !feet_only) {
        /* the motor-axis thread (geom 20) against the upper-leg cylinders: geom1 = the leg's cylinder (lower geom id), geom2 = the thread; the
         * normal points from the leg to the thread, the row is jac(motor body) - jac(upper leg).  MuJoCo's bounding-sphere filter first; then a
         * tight one of this oracle's own (the two AXES within r_thread + r_leg + 0.2 mm of each other - the HIP kernel's broad phase makes the same
         * test) that only spares the narrow phase where the answer is "apart". */
        CGeom th, c;
        cgeom_world(P, k, 20, &th);
        for (int l = 0; l < JB_NLEG; l++) {
            int g = 4 + 4 * l;
            cgeom_world(P, k, g, &c);
            double d[3] = {c.c[0] - th.c[0], c.c[1] - th.c[1], c.c[2] - th.c[2]};
            if (sqrt(dot3(d, d)) > cgeom_rbound(&th) + cgeom_rbound(&c)) continue;
            double axc[3] = {c.R[2], c.R[5], c.R[8]}, axt[3] = {th.R[2], th.R[5], th.R[8]};
            if (segment_distance(c.c, axc, c.sz[1], th.c, axt, th.sz[1]) >= th.sz[0] + c.sz[0] + 2e-4) { MARGIN(2e-4); continue; }
            double dist, dir[3], pos[3];
            pair_thread_geometric(&th, &c, JB_THREAD_BISECT, &dist, dir, pos);
            MARGIN(dist);
            if (dist < -c.sz[0]) MARGIN(0.0);          /* the leg's AXIS inside the thread: counted as ill-conditioned for the fp32 comparison, like the pair below */
            if (dist < 0) {
                if (n < MAXCON) {
                    con[n].dist = dist; memcpy(con[n].pos, pos, 24); memcpy(con[n].n, dir, 24);
                    con[n].body = (int)P[JB_P_GEOM + 20 * JB_GEOM_STRIDE + JB_G_BODY]; con[n].body2 = (int)P[JB_P_GEOM + g * JB_GEOM_STRIDE + JB_G_BODY];
                    con[n].geom = JB_NGEOM + 4 + l; con[n].slot = 0; con[n].wslot = WARM_PAIR2 + 4 * l;
                    n++;
                } else *overflow = 1;
            }
        }
    }
    if ((pair_contacts & 1) && !feet_only) {
        /* The geom pairs that can touch (on randomised models the eccentric-mass ellipsoid, geom 21 on the motor body,
         * reaches the upper-leg cylinders; every other pair MuJoCo's filters let through keeps its distance in every regime
         * measured).  MuJoCo orders a pair by geom type (ellipsoid < cylinder), so geom1 = the ellipsoid and the normal points from
         * the mass to the leg; bounding spheres first (mj_collideGeoms), then - where MuJoCo runs mjc_Convex = MPR - the geometric
         * contact MPR's output scatters around (pair_geometric above). */
        CGeom e, c;
        cgeom_world(P, k, 21, &e);
        for (int l = 0; l < JB_NLEG; l++) {
            int g = 4 + 4 * l;
            cgeom_world(P, k, g, &c);
            double d[3] = {c.c[0] - e.c[0], c.c[1] - e.c[1], c.c[2] - e.c[2]};
            if (sqrt(dot3(d, d)) > cgeom_rbound(&e) + cgeom_rbound(&c)) continue;
            {   /* a second, tight filter (not MuJoCo's; it only saves this oracle the narrow phase where the answer is "apart by more than
                 * 0.2 mm"): the cylinder's axis segment against the ellipsoid with semi-axes s + r_cyl + 0.2 mm, in that ellipsoid's unit-sphere
                 * coordinates - the same test the HIP kernel's broad phase makes */
                double cl[3], ul[3], ax[3] = {c.R[2], c.R[5], c.R[8]}, cs[3], us[3], p[3];
                matTvec3(cl, e.R, d);
                matTvec3(ul, e.R, ax);
                for (int i = 0; i < 3; i++) { double is = 1.0 / (e.sz[i] + c.sz[0] + 2e-4); cs[i] = cl[i] * is; us[i] = ul[i] * is; }
                double t = -dot3(cs, us) / dot3(us, us);
                t = t < -c.sz[1] ? -c.sz[1] : (t > c.sz[1] ? c.sz[1] : t);
                for (int i = 0; i < 3; i++) p[i] = cs[i] + t * us[i];
                if (dot3(p, p) >= 1.0) { MARGIN(2e-4); continue; }
            }
            double dist, dir[3], pos[3];
            pair_geometric(&e, &c, &dist, dir, pos);
            MARGIN(dist);
            /* An overlap deeper than the cylinder radius puts the leg's AXIS inside the mass: the nearest-surface point of an interior point
             * of this flat ellipsoid (3 mm half thickness) is ill-conditioned towards its mid-plane, and no robot could be built that
             * way - such env-steps count as ill-conditioned for the fp32 comparison (the contact itself is simulated all the same). */
            if (dist < -c.sz[0]) MARGIN(0.0);
            if (dist < 0) {
                if (n < MAXCON) {
                    con[n].dist = dist; memcpy(con[n].pos, pos, 24); memcpy(con[n].n, dir, 24);
                    con[n].body = (int)P[JB_P_GEOM + g * JB_GEOM_STRIDE + JB_G_BODY]; con[n].body2 = (int)P[JB_P_GEOM + 21 * JB_GEOM_STRIDE + JB_G_BODY];
                    con[n].geom = JB_NGEOM + l; con[n].slot = 0; con[n].wslot = WARM_PAIR + 4 * l;
                    n++;
                } else *overflow = 1;
            }
        }
    }
    return n;
}

/* MuJoCo's mju_makeFrame: complete a unit normal x to a right-handed frame (x, y, z) */
static void make_frame(const double* x, double* y, double* z) {
    if (x[1] < 0.5 && x[1] > -0.5) { y[0] = 0; y[1] = 1; y[2] = 0; } else { y[0] = 0; y[1] = 0; y[2] = 1; }
    double d = dot3(x, y);
    for (int i = 0; i < 3; i++) y[i] -= d * x[i];
    vnormalize(y);
    cross3(z, x, y);
}

/* impedance d(r) (MuJoCo solimp, 5-parameter form) */
static double impedance(const double* solimp, double r) {
    double d0 = solimp[0], dw = solimp[1], width = solimp[2], mid = solimp[3], pw = solimp[4];
    double x = fabs(r) / width, y;
    if (x >= 1) return dw;
    if (x <= 0) return d0;
    if (pw == 1) y = x;
    else if (x <= mid) y = pow(x / mid, pw) * mid;           /* a x^p, a = 1/mid^(p-1) */
    else y = 1 - pow((1 - x) / (1 - mid), pw) * (1 - mid);   /* 1 - b (1-x)^p        */
    return d0 + y * (dw - d0);
}

/* debug taps for the tests */
typedef struct {
    int ncon, nrow;
    double con_dist[MAXCON], con_pos[MAXCON][3];
    int con_geom[MAXCON];
    double f[MAXROW], aref[MAXROW], Rdiag[MAXROW], jar[MAXROW];   /* jar = J qacc - aref at the solution */
    double M[NV * NV], bias[NV], tau[NV], qacc_smooth[NV], qacc[NV], qfrc_constraint[NV];
} jbo_debug;

/* ------------------------------------------------------------------ one substep */
/* warm: optional [WARM_SIZE]: pyramid forces keyed by (geom, contact slot, edge), then the last qacc */
static void substep(const double* P, double* qpos, double* qvel, double ctrl, const jbo_opts* o,
                    double* warm, jbo_stats* st, jbo_debug* dbg) {
    Kin k;
    double M[NV * NV], L[NV * NV], bias[NV], tau[NV], qacc_s[NV], qfc[NV];
    const double h = P[JB_P_TIMESTEP];
    kinematics(P, qpos, &k);
    dynamics(P, &k, qvel, M, bias);

    /* passive (spring + damper, springref = 0) and actuator (general: gain*ctrl + bias) */
    for (int d = 0; d < NV; d++) tau[d] = -bias[d];
    for (int j = 0; j < NH; j++) {
        const double* H = P + JB_P_HINGE + j * JB_HINGE_STRIDE;
        tau[6 + j] += -H[JB_H_STIFFNESS] * qpos[7 + j] - H[JB_H_DAMPING] * qvel[6 + j];
    }
    double u = ctrl;
    if (u < P[JB_P_CTRLRANGE]) u = P[JB_P_CTRLRANGE];
    if (u > P[JB_P_CTRLRANGE + 1]) u = P[JB_P_CTRLRANGE + 1];
    {
        double gear = P[JB_P_GEAR], len = gear * qpos[15], vel = gear * qvel[14];
        double force = P[JB_P_GAIN] * u + P[JB_P_BIASPRM] + P[JB_P_BIASPRM + 1] * len + P[JB_P_BIASPRM + 2] * vel;
        tau[14] += gear * force;
    }

    memset(qfc, 0, sizeof qfc);
    Contact con[MAXCON];
    int ncon = 0, overflow = 0;
    double margin = INFINITY;
    if (o->contacts) ncon = collide(P, &k, o->feet_only, o->pair_contacts, con, &overflow, &margin);
    if (st && margin < st->margin_min) st->margin_min = margin;
    if (dbg) { dbg->ncon = ncon; dbg->nrow = 0; memcpy(dbg->M, M, sizeof M); memcpy(dbg->bias, bias, sizeof bias); memcpy(dbg->tau, tau, sizeof tau); }

    if (ncon > 0 || dbg) {
        memcpy(L, M, sizeof M);
        chol(L, NV, NV);
        memcpy(qacc_s, tau, sizeof tau);
        chol_solve(L, NV, NV, qacc_s);
        if (dbg) memcpy(dbg->qacc_smooth, qacc_s, sizeof qacc_s);
    }
    if (ncon > 0) {
        const int nr = 4 * ncon;
        const double mu = P[JB_P_FRICTION] * sqrt(1.0 / P[JB_P_IMPRATIO]);
        const double* solref = P + JB_P_SOLREF;
        const double* solimp = P + JB_P_SOLIMP;
        double tc = solref[0] < 2 * h ? 2 * h : solref[0];
        double dmax = solimp[1];
        double Kk = 1.0 / (dmax * dmax * tc * tc * solref[1] * solref[1]);
        double Bb = 2.0 / (dmax * tc);
        /* rows */
        double (*J)[NV] = malloc(sizeof(double) * NV * nr);
        double (*MJ)[NV] = malloc(sizeof(double) * NV * nr);
        double* A = malloc(sizeof(double) * nr * nr);
        double *b = malloc(sizeof(double) * nr), *Rr = malloc(sizeof(double) * nr), *f = malloc(sizeof(double) * nr), *aref = malloc(sizeof(double) * nr);
        for (int c = 0; c < ncon; c++) {
            double Jv[NV][3], Jw[NV][3];
            jacobian(&k, con[c].body, con[c].pos, Jv, Jw);
            /* contact frame: x = normal; MuJoCo mju_makeFrame -> for the floor (0,0,1): y = (0,1,0), z = x cross y = (-1,0,0) */
            double fx[3] = {con[c].n[0], con[c].n[1], con[c].n[2]}, fy[3], fz[3];
            make_frame(fx, fy, fz);
            double imp = impedance(solimp, con[c].dist);
            double tran = P[JB_P_BODY + con[c].body * JB_BODY_STRIDE + JB_B_INVW_TRAN];   /* + world body: 0 */
            if (con[c].body2 >= 0) {     /* geom-geom contact: relative motion of the two bodies (jac2 - jac1), both inverse weights */
                double Jv2[NV][3], Jw2[NV][3];
                jacobian(&k, con[c].body2, con[c].pos, Jv2, Jw2);
                for (int d = 0; d < NV; d++) for (int i = 0; i < 3; i++) Jv[d][i] -= Jv2[d][i];
                tran += P[JB_P_BODY + con[c].body2 * JB_BODY_STRIDE + JB_B_INVW_TRAN];
            }
            double fr = P[JB_P_FRICTION];
            double dA = tran + fr * fr * tran;
            double R0 = (1 - imp) / imp * dA;
            if (R0 < 1e-15) R0 = 1e-15;
            double Rpy = 2 * mu * mu * R0;
            for (int e = 0; e < 4; e++) {
                int r = 4 * c + e;
                double sgn = (e & 1) ? -1.0 : 1.0;
                double vel = 0;
                for (int d = 0; d < NV; d++) {
                    double jn = dot3(Jv[d], fx);
                    double jt = (e < 2) ? dot3(Jv[d], fy) : dot3(Jv[d], fz);
                    J[r][d] = jn + sgn * mu * jt;
                    vel += J[r][d] * qvel[d];
                }
                Rr[r] = Rpy;
                aref[r] = -Bb * vel - Kk * imp * con[c].dist;
                for (int d = 0; d < NV; d++) MJ[r][d] = J[r][d];
                chol_solve(L, NV, NV, MJ[r]);
                double a0 = 0;
                for (int d = 0; d < NV; d++) a0 += J[r][d] * qacc_s[d];
                b[r] = a0 - aref[r];
                f[r] = (warm && o->warmstart) ? warm[con[c].wslot + e] : 0.0;
            }
        }
        for (int i = 0; i < nr; i++)
            for (int j = 0; j < nr; j++) { double s = 0; for (int d = 0; d < NV; d++) s += J[i][d] * MJ[j][d]; A[i * nr + j] = s; }
        int sweeps = 0;
        double resid = 0;
        if (o->solver == 0) {
            /* projected Gauss-Seidel on the dual  min 1/2 f'(A+R)f + f'b, f>=0 */
            for (int it = 0; it < o->solver_iters; it++) {
                double maxd = 0;
                for (int i = 0; i < nr; i++) {
                    double s = b[i] + Rr[i] * f[i];
                    for (int j = 0; j < nr; j++) s += A[i * nr + j] * f[j];
                    double fn = f[i] - s / (A[i * nr + i] + Rr[i]);
                    if (fn < 0) fn = 0;
                    double dd = fabs(fn - f[i]) * (A[i * nr + i] + Rr[i]);   /* in acceleration units */
                    if (dd > maxd) maxd = dd;
                    f[i] = fn;
                }
                sweeps++;
                resid = maxd;
                if (o->solver_tol > 0 && maxd < o->solver_tol) break;
            }
        } else {
            /* primal Newton (MuJoCo's default solver):
             *   min_x 1/2 (x-xs)' M (x-xs) + sum_i 1/(2 R_i) min(0, J_i x - aref_i)^2
             * exact Hessian on the active set + exact line search on the
             * piecewise-quadratic cost.  Warm start: previous qacc (qacc_warmstart)
             * if its cost is lower than that of qacc_smooth. */
            double x[NV], g[NV], p[NV], H[NV * NV], Mp[NV];
            double *r = malloc(sizeof(double) * nr), *sj = malloc(sizeof(double) * nr), *bp = malloc(sizeof(double) * nr);
            int* idx = malloc(sizeof(int) * nr);
            memcpy(x, qacc_s, sizeof x);
            if (warm && o->warmstart) {
                double cw = 0, cs = 0, dx[NV];
                const double* xw = warm + JB_NGEOM * 16;
                for (int d = 0; d < NV; d++) dx[d] = xw[d] - qacc_s[d];
                for (int i = 0; i < NV; i++) for (int j = 0; j < NV; j++) cw += 0.5 * dx[i] * M[i * NV + j] * dx[j];
                for (int i = 0; i < nr; i++) {
                    double rw = -aref[i], rs = -aref[i];
                    for (int d = 0; d < NV; d++) { rw += J[i][d] * xw[d]; rs += J[i][d] * qacc_s[d]; }
                    if (rw < 0) cw += 0.5 * rw * rw / Rr[i];
                    if (rs < 0) cs += 0.5 * rs * rs / Rr[i];
                }
                if (cw < cs) memcpy(x, xw, sizeof x);
            }
            for (int it = 0; it < (o->solver_iters < 100 ? o->solver_iters : 100); it++) {
                /* gradient and Hessian */
                for (int i = 0; i < NV; i++) { double s = -tau[i]; for (int j = 0; j < NV; j++) s += M[i * NV + j] * x[j]; g[i] = s; }
                memcpy(H, M, sizeof H);
                for (int i = 0; i < nr; i++) {
                    double ri = -aref[i];
                    for (int d = 0; d < NV; d++) ri += J[i][d] * x[d];
                    r[i] = ri;
                    if (ri < 0) {
                        double Di = 1.0 / Rr[i];
                        for (int d = 0; d < NV; d++) {
                            g[d] += Di * ri * J[i][d];
                            for (int e2 = 0; e2 <= d; e2++) H[d * NV + e2] += Di * J[i][d] * J[i][e2];
                        }
                    }
                }
                for (int i = 0; i < NV; i++) for (int j = 0; j < i; j++) H[j * NV + i] = H[i * NV + j];
                chol(H, NV, NV);
                for (int d = 0; d < NV; d++) p[d] = -g[d];
                chol_solve(H, NV, NV, p);
                /* exact line search: phi'(a) = c0 + a c1 + sum_i D_i min(0, r_i + a s_i) s_i = 0 */
                double c0 = 0, c1 = 0;
                for (int i = 0; i < NV; i++) { double s = 0; for (int j = 0; j < NV; j++) s += M[i * NV + j] * p[j]; Mp[i] = s; }
                for (int i = 0; i < NV; i++) { double s = -tau[i]; for (int j = 0; j < NV; j++) s += M[i * NV + j] * x[j]; c0 += s * p[i]; c1 += p[i] * Mp[i]; }
                int nb = 0;
                double slope = c1, icpt = c0;        /* phi'(a) = icpt + slope a on the current segment */
                for (int i = 0; i < nr; i++) {
                    double si = 0; for (int d = 0; d < NV; d++) si += J[i][d] * p[d];
                    sj[i] = si;
                    if (r[i] < 0) { slope += si * si / Rr[i]; icpt += r[i] * si / Rr[i]; }
                    if ((r[i] < 0 && si > 0) || (r[i] >= 0 && si < 0)) { bp[nb] = -r[i] / si; idx[nb] = i; nb++; }
                }
                for (int i = 1; i < nb; i++) {      /* insertion sort of the breakpoints */
                    double a = bp[i]; int ii = idx[i], j = i - 1;
                    while (j >= 0 && bp[j] > a) { bp[j + 1] = bp[j]; idx[j + 1] = idx[j]; j--; }
                    bp[j + 1] = a; idx[j + 1] = ii;
                }
                double alpha = 0;
                int found = 0;
                for (int kbp = 0; kbp <= nb; kbp++) {
                    double a_hi = (kbp < nb) ? bp[kbp] : INFINITY;
                    double a0 = -icpt / slope;
                    if (a0 <= a_hi) { alpha = a0; found = 1; break; }
                    int i = idx[kbp];              /* row i toggles at a_hi */
                    /* row enters the active set when s<0 (residual decreasing), leaves when s>0 */
                    double Di = 1.0 / Rr[i], sgn;
                    sgn = (sj[i] < 0) ? 1.0 : -1.0;
                    slope += sgn * Di * sj[i] * sj[i];
                    icpt += sgn * Di * r[i] * sj[i];
                }
                if (!found) alpha = 1.0;
                if (o->solver == 2) alpha = 1.0;        /* experiment: semismooth Newton with full steps */
                double maxstep = 0;
                for (int d = 0; d < NV; d++) { double dx = alpha * p[d]; x[d] += dx; double sc = fabs(dx) * M[d * NV + d]; if (sc > maxstep) maxstep = sc; }
                sweeps++;
                resid = maxstep;
                if (maxstep <= o->solver_tol) break;
            }
            for (int i = 0; i < nr; i++) {
                double ri = -aref[i];
                for (int d = 0; d < NV; d++) ri += J[i][d] * x[d];
                f[i] = ri < 0 ? -ri / Rr[i] : 0.0;
            }
            if (warm) memcpy(warm + JB_NGEOM * 16, x, sizeof x);
            free(r); free(sj); free(bp); free(idx);
        }
        if (st) {
            st->sweeps_total += sweeps; if (sweeps > st->sweeps_max) st->sweeps_max = sweeps;
            st->nsolve++; if (resid > st->resid_max) st->resid_max = resid;
        }
        for (int d = 0; d < NV; d++) { double s = 0; for (int r = 0; r < nr; r++) s += J[r][d] * f[r]; qfc[d] = s; }
        if (warm) {
            memset(warm, 0, sizeof(double) * JB_NGEOM * 16);
            memset(warm + WARM_PAIR, 0, sizeof(double) * 32);
            for (int c = 0; c < ncon; c++) for (int e = 0; e < 4; e++) warm[con[c].wslot + e] = f[4 * c + e];
        }
        if (dbg) {
            dbg->nrow = nr;
            double qa[NV];
            for (int d = 0; d < NV; d++) qa[d] = tau[d] + qfc[d];
            chol_solve(L, NV, NV, qa);
            memcpy(dbg->qacc, qa, sizeof qa);
            for (int c = 0; c < ncon; c++) { dbg->con_dist[c] = con[c].dist; dbg->con_geom[c] = con[c].geom; memcpy(dbg->con_pos[c], con[c].pos, sizeof(double) * 3); }
            for (int r = 0; r < nr; r++) {
                double s = 0; for (int d = 0; d < NV; d++) s += J[r][d] * qa[d];
                dbg->f[r] = f[r]; dbg->aref[r] = aref[r]; dbg->Rdiag[r] = Rr[r]; dbg->jar[r] = s - aref[r];
            }
        }
        free(J); free(MJ); free(A); free(b); free(Rr); free(f); free(aref);
    } else if (warm) {
        memset(warm, 0, sizeof(double) * JB_NGEOM * 16);
        memset(warm + WARM_PAIR, 0, sizeof(double) * 32);
    }
    if (st) { st->ncon_last = ncon; if (ncon > st->ncon_max) st->ncon_max = ncon; if (overflow) st->overflow = 1; }
    if (dbg) memcpy(dbg->qfrc_constraint, qfc, sizeof qfc);

    /* Euler, implicit in joint damping:  (M + h diag(b)) qacc = qfrc_smooth + qfrc_constraint */
    double qacc[NV];
    memcpy(L, M, sizeof M);
    if (o->implicit_damp)
        for (int j = 0; j < NH; j++) L[(6 + j) * NV + 6 + j] += h * P[JB_P_HINGE + j * JB_HINGE_STRIDE + JB_H_DAMPING];
    chol(L, NV, NV);
    for (int d = 0; d < NV; d++) qacc[d] = tau[d] + qfc[d];
    chol_solve(L, NV, NV, qacc);
    if (dbg && ncon == 0) memcpy(dbg->qacc, qacc, sizeof qacc);

    /* mj_advance: velocity first, then position with the NEW velocity */
    for (int d = 0; d < NV; d++) qvel[d] += h * qacc[d];
    for (int i = 0; i < 3; i++) qpos[i] += h * qvel[i];
    {   /* mju_quatIntegrate: q <- q * exp(h w / 2), w body-local */
        double w[3] = {qvel[3], qvel[4], qvel[5]};
        double nrm = sqrt(dot3(w, w));
        if (nrm > 1e-15) {
            double ang = h * nrm, s = sin(0.5 * ang) / nrm, c = cos(0.5 * ang);
            double dq[4] = {c, w[0] * s, w[1] * s, w[2] * s}, *q = qpos + 3;
            double r0 = q[0] * dq[0] - q[1] * dq[1] - q[2] * dq[2] - q[3] * dq[3];
            double r1 = q[0] * dq[1] + q[1] * dq[0] + q[2] * dq[3] - q[3] * dq[2];
            double r2 = q[0] * dq[2] - q[1] * dq[3] + q[2] * dq[0] + q[3] * dq[1];
            double r3 = q[0] * dq[3] + q[1] * dq[2] - q[2] * dq[1] + q[3] * dq[0];
            double n = sqrt(r0 * r0 + r1 * r1 + r2 * r2 + r3 * r3);
            q[0] = r0 / n; q[1] = r1 / n; q[2] = r2 / n; q[3] = r3 / n;
        }
    }
    for (int j = 0; j < NH; j++) qpos[7 + j] += h * qvel[6 + j];
}

/* ------------------------------------------------------------------ public: physics */
void jbo_default_opts(jbo_opts* o) {
    o->contacts = 1; o->implicit_damp = 1; o->solver_iters = 20000; o->solver_tol = 1e-12; o->warmstart = 1; o->feet_only = 0; o->solver = 1; o->pair_contacts = 3;
}

/* nsub substeps with constant ctrl (reference: control.Environment.step, 50 substeps) */
void jbo_step_physics(const double* P, double* qpos, double* qvel, double ctrl, int nsub, const jbo_opts* o,
                      double* warm, jbo_stats* st) {
    for (int s = 0; s < nsub; s++) substep(P, qpos, qvel, ctrl, o, warm, st, NULL);
    /* trailing mj_step1: normalise quaternion so derived quantities match the state */
    double* q = qpos + 3;
    double n = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2] + q[3] * q[3]);
    for (int i = 0; i < 4; i++) q[i] /= n;
}

/* forward dynamics only, with taps (qpos/qvel not advanced) */
void jbo_forward_debug(const double* P, const double* qpos, const double* qvel, double ctrl, const jbo_opts* o, jbo_debug* dbg) {
    double q[NQ], v[NV];
    memcpy(q, qpos, sizeof q); memcpy(v, qvel, sizeof v);
    substep(P, q, v, ctrl, o, NULL, NULL, dbg);
}
int jbo_debug_size(void) { return (int)sizeof(jbo_debug); }
/* the narrow phase of the mass-ellipsoid / upper-leg-cylinder pair of leg l in one configuration, for the tests:
 * out = [depth, dir(3) from the ellipsoid to the cylinder, pos(3)]; returns 1 when the two intersect, 0 otherwise (out[0] = GJK gap) */
int jbo_pair_mpr(const double* P, const double* qpos_in, int leg, double tol, int iters, double* out) {
    Kin k; double qpos[NQ];
    memcpy(qpos, qpos_in, sizeof qpos);
    kinematics(P, qpos, &k);
    CGeom e, c;
    cgeom_world(P, &k, 21, &e);
    cgeom_world(P, &k, 4 + 4 * leg, &c);
    g_mpr_tol = tol > 0 ? tol : MPR_TOL; g_mpr_iters = iters > 0 ? iters : MPR_ITERS;
    int hit = mpr_penetration(&e, &c, out, out + 1, out + 4) == 0;
    g_mpr_tol = MPR_TOL; g_mpr_iters = MPR_ITERS;
    if (!hit) out[0] = jbo_posed_geom_distance(e.type, e.c, e.R, e.sz, c.type, c.c, c.R, c.sz);
    return hit;
}
/* ... and the geometric contact of the same pair (what collide() uses): out = [dist (<0: penetration), n(3), pos(3)]; 0 if undefined */
int jbo_pair_geometric_exact(const double* P, const double* qpos_in, int leg, double* out) {
    Kin k; double qpos[NQ];
    memcpy(qpos, qpos_in, sizeof qpos);
    kinematics(P, qpos, &k);
    CGeom e, c;
    cgeom_world(P, &k, 21, &e);
    cgeom_world(P, &k, 4 + 4 * leg, &c);
    return pair_geometric_exact(&e, &c, out, out + 1, out + 4);
}
/* the thread (geom 20) against the upper-leg cylinder of `leg`: out = [dist, n (leg -> thread, 3), pos (3), distance of the two axis segments] */
int jbo_pair_thread_geometric(const double* P, const double* qpos_in, int leg, int steps, double* out) {
    Kin k; double qpos[NQ];
    memcpy(qpos, qpos_in, sizeof qpos);
    kinematics(P, qpos, &k);
    CGeom th, c;
    cgeom_world(P, &k, 20, &th);
    cgeom_world(P, &k, 4 + 4 * leg, &c);
    pair_thread_geometric(&th, &c, steps > 0 ? steps : JB_THREAD_BISECT, out, out + 1, out + 4);
    double axc[3] = {c.R[2], c.R[5], c.R[8]}, axt[3] = {th.R[2], th.R[5], th.R[8]};
    out[7] = segment_distance(c.c, axc, c.sz[1], th.c, axt, th.sz[1]);
    return 1;
}
int jbo_pair_geometric(const double* P, const double* qpos_in, int leg, double* out) {
    Kin k; double qpos[NQ];
    memcpy(qpos, qpos_in, sizeof qpos);
    kinematics(P, qpos, &k);
    CGeom e, c;
    cgeom_world(P, &k, 21, &e);
    cgeom_world(P, &k, 4 + 4 * leg, &c);
    return pair_geometric(&e, &c, out, out + 1, out + 4);
}

/* total momentum (linear P, angular L about world origin) and energies, for the conservation tests */
void jbo_momentum_energy(const double* P, const double* qpos_in, const double* qvel, double* out /*[8]: P3 L3 T V*/) {
    Kin k; double qpos[NQ];
    memcpy(qpos, qpos_in, sizeof qpos);
    kinematics(P, qpos, &k);
    double lin[3] = {0, 0, 0}, ang[3] = {0, 0, 0}, T = 0, V = 0;
    for (int b = 0; b < NB; b++) {
        double Jv[NV][3], Jw[NV][3], vc[3] = {0, 0, 0}, w[3] = {0, 0, 0}, Iw_[3], t[3];
        jacobian(&k, b, k.com[b], Jv, Jw);
        for (int d = 0; d < NV; d++) for (int i = 0; i < 3; i++) { vc[i] += Jv[d][i] * qvel[d]; w[i] += Jw[d][i] * qvel[d]; }
        matvec3(Iw_, k.Iw[b], w);
        cross3(t, k.com[b], vc);
        for (int i = 0; i < 3; i++) { lin[i] += k.mass[b] * vc[i]; ang[i] += Iw_[i] + k.mass[b] * t[i]; }
        T += 0.5 * k.mass[b] * dot3(vc, vc) + 0.5 * dot3(w, Iw_);
        V -= k.mass[b] * dot3(P + JB_P_GRAVITY, k.com[b]);
    }
    for (int j = 0; j < NH; j++) V += 0.5 * P[JB_P_HINGE + j * JB_HINGE_STRIDE + JB_H_STIFFNESS] * qpos[7 + j] * qpos[7 + j];
    for (int i = 0; i < 3; i++) { out[i] = lin[i]; out[3 + i] = ang[i]; }
    out[6] = T; out[7] = V;
}

/* ------------------------------------------------------------------ RNG: Philox4x32-10 */
static inline void philox_round(uint32_t* c, const uint32_t* k) {
    uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k[0], n1 = (uint32_t)p1;
    uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k[1], n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
}
void jbo_philox(uint64_t seed, uint64_t env, uint32_t episode, uint32_t stream, uint32_t out[4]) {
    uint32_t c[4] = {(uint32_t)env, (uint32_t)(env >> 32), episode, stream};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    for (int r = 0; r < 10; r++) {
        philox_round(c, k);
        k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
    }
    memcpy(out, c, sizeof(uint32_t) * 4);
}
static inline double u01(uint32_t x) { return (double)(x >> 8) * (1.0 / 16777216.0); }   /* [0,1), 24 bits: exact in fp32 too */

/* ------------------------------------------------------------------ reset (reference jitterbug.py:601-666) */
void jbo_reset(const double* P, int task, int random_pose, uint64_t seed, uint64_t env, uint32_t episode,
               double* qpos, double* qvel, double* target /*[3]: tx ty psi*/) {
    uint32_t r0[4], r1[4];
    jbo_philox(seed, env, episode, 0, r0);
    jbo_philox(seed, env, episode, 1, r1);
    const double TWO_PI = 2.0 * M_PI;
    double angle = u01(r0[0]) * TWO_PI;              /* :609 */
    double radius = 0.05 + u01(r0[1]) * (0.2 - 0.05); /* :610 */
    double yaw = u01(r0[2]) * TWO_PI;                /* :611 */
    memset(qpos, 0, sizeof(double) * NQ);
    memset(qvel, 0, sizeof(double) * NV);
    for (int i = 0; i < 3; i++) qpos[i] = P[JB_P_ROOTPOS0 + i];
    qpos[3] = 1.0;
    target[0] = target[1] = target[2] = 0.0;
    if (task == JB_TASK_FACE_DIRECTION || task == JB_TASK_MOVE_IN_DIRECTION) target[2] = yaw;       /* :618-630 */
    else if (task == JB_TASK_MOVE_TO_POSITION) { target[0] = radius * cos(angle); target[1] = radius * sin(angle); } /* :632-639 */
    else if (task == JB_TASK_MOVE_TO_POSE) { target[0] = radius * cos(angle); target[1] = radius * sin(angle); target[2] = yaw; } /* :641-648 */
    if (random_pose) {                                /* :653-664 */
        double th = u01(r0[3]) * TWO_PI;
        double ax = u01(r1[0]) * 0.05 - 0.025, ay = u01(r1[1]) * 0.05 - 0.025, az = 1.0;
        double n = sqrt(ax * ax + ay * ay + az * az);
        double s = sin(0.5 * th);
        qpos[3] = cos(0.5 * th); qpos[4] = s * ax / n; qpos[5] = s * ay / n; qpos[6] = s * az / n;
    }
}

/* ------------------------------------------------------------------ observation / reward */
static double wrap_pi(double a) {           /* reference jitterbug.py:235-238, 313-316: (-pi, pi] */
    while (a > M_PI) a -= 2 * M_PI;
    while (a <= -M_PI) a += 2 * M_PI;
    return a;
}
static inline double norm_(double v, double lo, double hi) { return (v - lo) / (hi - lo) * 2.0 - 1.0; }   /* :668-671 */

static void target_quat(double psi, double* q) { q[0] = cos(0.5 * psi); q[1] = 0; q[2] = 0; q[3] = sin(0.5 * psi); }

static double angle_to_target(const double* qpos, double psi) {   /* :192-208, 262-273, 305-317 */
    double R[9], Rt[9], tq[4];
    quat2mat(R, qpos + 3);
    double yaw = atan2(R[3], R[0]) - M_PI / 2;
    target_quat(psi, tq);
    quat2mat(Rt, tq);
    double tyaw = atan2(Rt[3], Rt[0]);
    return wrap_pi(tyaw - yaw);
}
static void target_in_jb_frame(const double* P, const double* qpos, const double* target, double* out) {   /* :275-290 */
    double R[9], d[3] = {target[0] - qpos[0], target[1] - qpos[1], P[JB_P_TARGETZ] - qpos[2]};
    quat2mat(R, qpos + 3);
    matTvec3(out, R, d);
}
/* sensor jitterbug_framelinvel (reference jitterbug.xml:121, objtype="body"): MuJoCo's mj_objectVelocity takes an mjOBJ_BODY
 * object at the body's inertial frame (xipos = the root body's own COM), world axes:  v + R (w_body x ipos). */
void jbo_framelinvel(const double* P, const double* qpos, const double* qvel, double* out) {
    double R[9], l[3], lw[3];
    quat2mat(R, qpos + 3);
    cross3(l, qvel + 3, P + JB_P_BODY + JB_B_COM);
    matvec3(lw, R, l);
    for (int i = 0; i < 3; i++) out[i] = qvel[i] + lw[i];
}
static void vel_in_target_frame(const double* P, const double* qpos, const double* qvel, double psi, double* out) {   /* :292-303 */
    double Rt[9], tq[4], v[3];
    target_quat(psi, tq);
    quat2mat(Rt, tq);
    jbo_framelinvel(P, qpos, qvel, v);
    matTvec3(out, Rt, v);
}

int jbo_obs_dim(int task) { static const int d[JB_NTASK] = {15, 16, 19, 18, 19}; return (task >= 0 && task < JB_NTASK) ? d[task] : -1; }

/* reference jitterbug.py:673-763 (un-encoded observation dict, flattened in dict order) */
void jbo_observation(const double* P, int task, const double* qpos, const double* qvel, const double* target, double* obs) {
    obs[0] = norm_(qpos[0], -2, 2); obs[1] = norm_(qpos[1], -2, 2); obs[2] = norm_(qpos[2], 0, 0.1);
    for (int i = 0; i < 4; i++) obs[3 + i] = norm_(qpos[3 + i], -1, 1);
    for (int i = 0; i < 3; i++) obs[7 + i] = norm_(qvel[i], -1, 1);
    for (int i = 0; i < 3; i++) obs[10 + i] = norm_(qvel[3 + i], -35, 35);
    obs[13] = norm_(wrap_pi(qpos[15] + M_PI / 2), -M_PI, M_PI);       /* :222-239 */
    obs[14] = norm_(qvel[14], -180, 180);
    double t3[3];
    switch (task) {
    case JB_TASK_MOVE_FROM_ORIGIN: break;
    case JB_TASK_FACE_DIRECTION:
        obs[15] = norm_(angle_to_target(qpos, target[2]), -M_PI, M_PI); break;
    case JB_TASK_MOVE_IN_DIRECTION:
        obs[15] = norm_(angle_to_target(qpos, target[2]), -M_PI, M_PI);
        vel_in_target_frame(P, qpos, qvel, target[2], t3);
        for (int i = 0; i < 3; i++) obs[16 + i] = norm_(t3[i], -1, 1);
        break;
    case JB_TASK_MOVE_TO_POSITION:
        target_in_jb_frame(P, qpos, target, t3);
        obs[15] = norm_(t3[0], -3, 3); obs[16] = norm_(t3[1], -3, 3); obs[17] = norm_(t3[2], -0.1, 0.1); break;
    case JB_TASK_MOVE_TO_POSE:
        target_in_jb_frame(P, qpos, target, t3);
        obs[15] = norm_(t3[0], -3, 3); obs[16] = norm_(t3[1], -3, 3); obs[17] = norm_(t3[2], -0.1, 0.1);
        obs[18] = norm_(angle_to_target(qpos, target[2]), -M_PI, M_PI); break;
    }
}

/* dm_control utils/rewards.tolerance restated (third-party; reference call sites jitterbug.py:846-889) */
static double sigmoid_(double x, double value_at_1, int kind /*0 gaussian 1 cosine 2 linear*/) {
    if (kind == 0) { double scale = sqrt(-2 * log(value_at_1)); return exp(-0.5 * (x * scale) * (x * scale)); }
    if (kind == 1) { double scale = acos(2 * value_at_1 - 1) / M_PI, sx = x * scale; return fabs(sx) < 1 ? (1 + cos(M_PI * sx)) / 2 : 0.0; }
    { double scale = 1 - value_at_1, sx = x * scale; return fabs(sx) < 1 ? 1 - sx : 0.0; }
}
static double tolerance(double x, double lo, double hi, double margin, double value_at_margin, int kind) {
    int in_bounds = (lo <= x) && (x <= hi);
    if (margin == 0) return in_bounds ? 1.0 : 0.0;
    double d = (x < lo ? lo - x : x - hi) / margin;
    return in_bounds ? 1.0 : sigmoid_(d, value_at_margin, kind);
}

double jbo_reward(const double* P, int task, const double* qpos, const double* qvel, const double* target) {
    double R[9], t3[3], r = 0;
    quat2mat(R, qpos + 3);
    double upright = tolerance(R[8], 1, 1, 0.5, 0.1, 0);                                   /* :882-889 */
    double pos_r = 0, head_r = 0, vel_r = 0;
    target_in_jb_frame(P, qpos, target, t3);
    pos_r = tolerance(sqrt(dot3(t3, t3)), 0, 0, 0.05, 0.1, 0);                             /* :868-880 */
    head_r = tolerance(angle_to_target(qpos, target[2]), 0, 0, M_PI / 2, 0.0, 1);          /* :840-852 */
    vel_in_target_frame(P, qpos, qvel, target[2], t3);
    vel_r = tolerance(t3[0], 0.1, INFINITY, 0.1, 0.0, 2);                                  /* :854-866 */
    switch (task) {                                                                        /* :891-925 */
    case JB_TASK_MOVE_FROM_ORIGIN: r = 1 - pos_r; break;
    case JB_TASK_FACE_DIRECTION: r = head_r; break;
    case JB_TASK_MOVE_IN_DIRECTION: r = vel_r; break;
    case JB_TASK_MOVE_TO_POSITION: r = pos_r; break;
    case JB_TASK_MOVE_TO_POSE: r = pos_r * head_r; break;
    }
    return r * upright;
}
void jbo_reward_terms(const double* P, const double* qpos, const double* qvel, const double* target, double* out /*P,H,V,U*/) {
    double R[9], t3[3];
    quat2mat(R, qpos + 3);
    target_in_jb_frame(P, qpos, target, t3);
    out[0] = tolerance(sqrt(dot3(t3, t3)), 0, 0, 0.05, 0.1, 0);
    out[1] = tolerance(angle_to_target(qpos, target[2]), 0, 0, M_PI / 2, 0.0, 1);
    vel_in_target_frame(P, qpos, qvel, target[2], t3);
    out[2] = tolerance(t3[0], 0.1, INFINITY, 0.1, 0.0, 2);
    out[3] = tolerance(R[8], 1, 1, 0.5, 0.1, 0);
}
double jbo_tolerance(double x, double lo, double hi, double margin, double vam, int kind) { return tolerance(x, lo, hi, margin, vam, kind); }

/* ------------------------------------------------------------------ batch environment (CPU baseline + lockstep oracle) */
typedef struct {
    int n, task, random_pose, nsub, step_limit, per_env_model;
    uint64_t seed, env_offset;
    jbo_opts opts;
    double* P;          /* [NPARAM] or [n, NPARAM] */
    double *qpos, *qvel, *target, *warm;
    int *step_count; uint32_t* episode;
    double* margin;      /* [n]: jbo_stats.margin_min of each env's last control step */
    jbo_stats stats;
} jbo_env;

static const double* envP(const jbo_env* e, int i) { return e->per_env_model ? e->P + (size_t)i * JB_NPARAM : e->P; }

jbo_env* jbo_env_create(int n, int task, int random_pose, int nsub, int step_limit, uint64_t seed, uint64_t env_offset,
                        const double* P, int per_env_model, const jbo_opts* opts) {
    jbo_env* e = calloc(1, sizeof *e);
    e->n = n; e->task = task; e->random_pose = random_pose; e->nsub = nsub; e->step_limit = step_limit;
    e->seed = seed; e->env_offset = env_offset; e->per_env_model = per_env_model; e->opts = *opts;
    size_t np = per_env_model ? (size_t)n * JB_NPARAM : JB_NPARAM;
    e->P = malloc(sizeof(double) * np); memcpy(e->P, P, sizeof(double) * np);
    e->qpos = calloc((size_t)n * NQ, sizeof(double)); e->qvel = calloc((size_t)n * NV, sizeof(double));
    e->target = calloc((size_t)n * 3, sizeof(double)); e->warm = calloc((size_t)n * WARM_SIZE, sizeof(double));
    e->step_count = calloc(n, sizeof(int)); e->episode = calloc(n, sizeof(uint32_t)); e->margin = calloc(n, sizeof(double));
    /* like jb_create: a created env is already in a valid state (reset #0); the first explicit reset is #1 */
    for (int i = 0; i < n; i++) { jbo_reset(envP(e, i), e->task, e->random_pose, e->seed, e->env_offset + (uint64_t)i, 0, e->qpos + (size_t)i * NQ, e->qvel + (size_t)i * NV, e->target + (size_t)i * 3); e->episode[i] = 1; }
    return e;
}
void jbo_env_destroy(jbo_env* e) {
    if (!e) return;
    free(e->P); free(e->qpos); free(e->qvel); free(e->target); free(e->warm); free(e->step_count); free(e->episode); free(e->margin); free(e);
}
static void env_reset_one(jbo_env* e, int i) {
    jbo_reset(envP(e, i), e->task, e->random_pose, e->seed, e->env_offset + (uint64_t)i, e->episode[i],
              e->qpos + (size_t)i * NQ, e->qvel + (size_t)i * NV, e->target + (size_t)i * 3);
    memset(e->warm + (size_t)i * WARM_SIZE, 0, sizeof(double) * WARM_SIZE);
    e->step_count[i] = 0;
}
/* mask nullable (= all). Each reset consumes one episode index. obs nullable. */
void jbo_env_reset(jbo_env* e, const uint8_t* mask, double* obs) {
    int D = jbo_obs_dim(e->task);
    for (int i = 0; i < e->n; i++) {
        if (mask && !mask[i]) continue;
        env_reset_one(e, i);
        e->episode[i]++;
    }
    if (obs) for (int i = 0; i < e->n; i++)
        jbo_observation(envP(e, i), e->task, e->qpos + (size_t)i * NQ, e->qvel + (size_t)i * NV, e->target + (size_t)i * 3, obs + (size_t)i * D);
}
/* one control step for every env; auto_reset: VecEnv semantics (obs of the new episode is returned on done) */
void jbo_env_step(jbo_env* e, const double* action, double* obs, double* reward, uint8_t* done, int auto_reset, int nthreads) {
    int D = jbo_obs_dim(e->task);
    jbo_stats agg; memset(&agg, 0, sizeof agg);
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#pragma omp parallel
#endif
    {
        jbo_stats st; memset(&st, 0, sizeof st);
#ifdef _OPENMP
#pragma omp for schedule(dynamic, 8)
#endif
        for (int i = 0; i < e->n; i++) {
            double* qp = e->qpos + (size_t)i * NQ; double* qv = e->qvel + (size_t)i * NV; double* tg = e->target + (size_t)i * 3;
            st.margin_min = INFINITY;
            jbo_step_physics(envP(e, i), qp, qv, action[i], e->nsub, &e->opts, e->warm + (size_t)i * WARM_SIZE, &st);
            e->margin[i] = st.margin_min;
            e->step_count[i]++;
            if (reward) reward[i] = jbo_reward(envP(e, i), e->task, qp, qv, tg);
            int d = e->step_count[i] >= e->step_limit;
            if (done) done[i] = (uint8_t)d;
            if (d && auto_reset) { env_reset_one(e, i); e->episode[i]++; }
            if (obs) jbo_observation(envP(e, i), e->task, qp, qv, tg, obs + (size_t)i * D);
        }
#ifdef _OPENMP
#pragma omp critical
#endif
        {
            agg.sweeps_total += st.sweeps_total; agg.nsolve += st.nsolve;
            if (st.sweeps_max > agg.sweeps_max) agg.sweeps_max = st.sweeps_max;
            if (st.ncon_max > agg.ncon_max) agg.ncon_max = st.ncon_max;
            if (st.resid_max > agg.resid_max) agg.resid_max = st.resid_max;
            agg.overflow |= st.overflow;
        }
    }
    e->stats = agg;
}
void jbo_env_get_state(const jbo_env* e, double* qpos, double* qvel, double* target) {
    if (qpos) memcpy(qpos, e->qpos, sizeof(double) * (size_t)e->n * NQ);
    if (qvel) memcpy(qvel, e->qvel, sizeof(double) * (size_t)e->n * NV);
    if (target) memcpy(target, e->target, sizeof(double) * (size_t)e->n * 3);
}
void jbo_env_set_state(jbo_env* e, const double* qpos, const double* qvel, const double* target) {
    if (qpos) memcpy(e->qpos, qpos, sizeof(double) * (size_t)e->n * NQ);
    if (qvel) memcpy(e->qvel, qvel, sizeof(double) * (size_t)e->n * NV);
    if (target) memcpy(e->target, target, sizeof(double) * (size_t)e->n * 3);
    memset(e->warm, 0, sizeof(double) * (size_t)e->n * WARM_SIZE);
}
void jbo_env_get_counters(const jbo_env* e, int* step_count, uint32_t* episode) {
    if (step_count) memcpy(step_count, e->step_count, sizeof(int) * e->n);
    if (episode) memcpy(episode, e->episode, sizeof(uint32_t) * e->n);
}
void jbo_env_stats(const jbo_env* e, jbo_stats* out) { *out = e->stats; }
void jbo_env_get_margin(const jbo_env* e, double* out) { memcpy(out, e->margin, sizeof(double) * e->n); }
int jbo_max_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

int jbo_warm_size(void) { return WARM_SIZE; }
/* size of the OpenMP team for every later parallel region of this library (callers pass the cores they may really use) */
void jbo_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

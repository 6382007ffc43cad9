// CPU harness for the device arithmetic headers (fp29.cuh / ec.cuh are host+device code).  Built with
// g++ -DARKBP_CHECK_BOUNDS so every limb/value contract is asserted while tests/test_fp29_host.py
// compares results with Python integers.  This is a unit test of the kernels' arithmetic, not a
// product path: nothing in the library routes work through it.
#include "ec.cuh"
#include "ecq.cuh"
#include "glv.cuh"
using namespace arkbp;

template <class F> static void fe_op(int op, const u32* a, const u32* b, u32* out) {
    Fe x = fe_load_ark<F>(a), y = fe_load_ark<F>(b), r;
    switch (op) {
        case 0: r = fe_mul<F>(x, y); break;
        case 1: r = fe_norm(fe_add(x, y)); break;
        case 2: r = fe_sub<F, 2>(x, y); break;
        case 3: r = fe_sqr<F>(x); break;
        case 4: r = fe_inv<F>(x); break;
        case 5: r = x; break;
        case 6: fe_store_canon<F>(out, x); return;
        case 7: r = fe_neg<F, 2>(x); break;
        case 8: r = fe_mul<F>(fe_add(x, y), fe_add(x, x)); break;             // lazy operands (L = 2 each)
        case 9: r = fe_wred<F>(fe_sub<F, 16>(fe_sub<F, 8>(x, y), fe_times<8>(y))); break;  // big V, then weak-reduce
        case 10: { Fe c = fe_load_canon<F>(a); r = c; break; }
        case 11: { u32 t[8]; fe_store_dev<F>(t, x); r = fe_load_dev<F>(t); break; }
        case 12: out[0] = fe_is_zero_mod<F>(fe_sub<F, 4>(x, y)); return;
        case 13: r = fe_mul2<F>(x, y, fe_add(x, y), y); break;                                   // x*y + (x+y)*y, one reduction (L products 1 + 2)
        case 14: r = fe_mul2<F>(fe_add(x, x), y, fe_sub<F, 4>(y, x), fe_add(x, fe_add(y, y))); break;   // 2x*y + (y-x)*(x+2y): L products 2 + 3, V up to 5 * 3
        default: r = fe_zero<F>();
    }
    fe_store_ark<F>(out, r);
}
template <class C> static void pt_op(int op, const u32* p, const u32* q, const u32* k, u32* out) {
    typedef typename C::Fq F;
    Aff P = aff_load_ark<C>(p), Q = aff_load_ark<C>(q);
    Jac r;
    switch (op) {
        case 0: r = jac_add<C>(jac_from_aff<C>(P), jac_from_aff<C>(Q)); break;
        case 1: r = jac_madd<C>(jac_from_aff<C>(P), Q); break;
        case 2: r = jac_dbl<C>(jac_from_aff<C>(P)); break;
        case 3: {  // k*P (k: 8 canonical words), then + Q through the general adder with Z != 1 on both sides
            r = jac_inf<C>();
            for (int i = 255; i >= 0; i--) {
                r = jac_dbl<C>(r);
                if ((k[i >> 5] >> (i & 31)) & 1) r = jac_madd<C>(r, P);
            }
            Jac q2 = jac_dbl<C>(jac_from_aff<C>(Q));
            r = jac_add<C>(r, q2);
            break;
        }
        case 4: r = jac_madd<C>(jac_dbl<C>(jac_from_aff<C>(P)), aff_cneg_lazy<C>(Q, true)); break;  // 2P - Q
        case 5: { Aff n = aff_neg<C>(P); aff_store_ark<C>(out, n); return; }
        default: r = jac_inf<C>();
    }
    (void)sizeof(F);
    aff_store_ark<C>(out, jac_to_aff<C>(r));
}

// the quad-cooperative schedules (ecq.cuh) through the CPU stand-in for the DPP exchange: 16 rounds of the four lanes, then all four
// lanes must hold the lane-per-operation result.  op 0 add, 1 mixed add, 2 doubling; returns 0 when every lane agrees with ec.cuh.
template <class C> static int quad_op(int op, const u32* p, const u32* q, u32* out) {
    typedef typename C::Fq F;
    const Aff P = aff_load_ark<C>(p), Q = aff_load_ark<C>(q);
    // operands with Z != 1 for the Jacobian sides
    const Jac Pj = aff_is_inf(P) ? jac_inf<C>() : jac_dbl<C>(jac_madd<C>(jac_dbl<C>(jac_from_aff<C>(P)), aff_cneg_lazy<C>(P, true)));   // 2*(2P - P) = 2P, Z != 1
    const Jac Qj = aff_is_inf(Q) ? jac_inf<C>() : jac_madd<C>(jac_dbl<C>(jac_from_aff<C>(Q)), aff_cneg_lazy<C>(Q, true));               // 2Q - Q = Q, Z != 1
    const Jac want = op == 0 ? jac_add<C>(Pj, Qj) : op == 1 ? jac_madd<C>(Pj, Q) : jac_dbl<C>(Pj);
    Jac got[4];
    QuadSim& sim = quad_sim();
    sim = QuadSim();
    for (int round = 0; round < 16; round++)
        for (int lane = 0; lane < 4; lane++) {
            sim.site = 0; sim.lane = lane;
            got[lane] = op == 0 ? qjac_add<C>(Pj, Qj, (u32)lane) : op == 1 ? qjac_madd<C>(Pj, Q, (u32)lane) : qjac_dbl<C>(Pj, (u32)lane);
        }
    const Aff w = jac_to_aff<C>(want);
    int bad = 0;
    for (int lane = 0; lane < 4; lane++) {
        const Aff g = jac_to_aff<C>(got[lane]);
        if (!fe_eq_exact(g.x, w.x) || !fe_eq_exact(g.y, w.y)) bad |= 1 << lane;
    }
    aff_store_ark<C>(out, w);
    (void)sizeof(F);
    return bad;
}

extern "C" {
int fp29_quad_op(int cid, int op, const u32* p, const u32* q, u32* out) { return cid == 0 ? quad_op<Secq>(op, p, q, out) : quad_op<Zorro>(op, p, q, out); }
// glv_split of a canonical secq256k1 scalar (8 words) -> mag1[4] | mag2[4] | signs; returns 1 when both halves fit 128 bits
int fp29_glv_split(const u32* k, u32* out12) { return glv_split<Secq>(k, out12) ? 1 : 0; }
void fp29_fe_op(int fid, int op, const u32* a, const u32* b, u32* out) {
    switch (fid) {
        case 0: fe_op<SecqFq>(op, a, b, out); break;
        case 1: fe_op<SecqFr>(op, a, b, out); break;
        case 2: fe_op<ZorroFq>(op, a, b, out); break;
        default: fe_op<ZorroFr>(op, a, b, out); break;
    }
}
void fp29_pt_op(int cid, int op, const u32* p, const u32* q, const u32* k, u32* out) {
    if (cid == 0) pt_op<Secq>(op, p, q, k, out); else pt_op<Zorro>(op, p, q, k, out);
}
}

// CPU harness for the device arithmetic headers (fp29.cuh / ec.cuh are host+device code).  Built with
// g++ -DARKBP_CHECK_BOUNDS so every limb/value contract is asserted while tests/test_fp29_host.py
// compares results with Python integers.  This is a unit test of the kernels' arithmetic, not a
// product path: nothing in the library routes work through it.
#include "ec.cuh"
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

extern "C" {
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

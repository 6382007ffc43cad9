"""ctypes binding of libarkbp_hip.so (include/arkbp.h).  Fails loudly when the HIP library is missing
or no GPU is visible: there is no CPU fallback in the product path."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("ARKBP_LIB_PATH") or os.path.join(_HERE, "libarkbp_hip.so")   # (the override: sanitizer builds of the same sources, tools/sanitize/)
_lib = None

BP_OK, BP_E_ARG, BP_E_HIP, BP_E_NO_DEVICE, BP_E_VERIFICATION, BP_E_GENS_LENGTH, BP_E_FORMAT, BP_E_MISSING = 0, -1, -2, -3, -4, -5, -6, -7

# every symbol include/arkbp.h declares (tests check the library exports all of them)
EXPORTS = [
    "bp_last_error", "bp_device_count", "bp_ctx_create", "bp_ctx_destroy", "bp_ctx_sync",
    "bp_dev_alloc", "bp_dev_free", "bp_dev_upload", "bp_dev_download", "bp_points_import", "bp_points_export",
    "bp_msm", "bp_msm_gens", "bp_msm_dev", "bp_msm_window_count", "bp_msm_dev_windows", "bp_ipa_create", "bp_ipa_begin", "bp_ipa_round_LR", "bp_ipa_round_fold", "bp_ipa_finish", "bp_ipa_export", "bp_ipa_verify", "bp_gens_derive", "bp_gens_upload", "bp_gens_download", "bp_pedersen_gens",
    "bp_host_derive_generators", "bp_transcript_new", "bp_transcript_free", "bp_transcript_append_message", "bp_transcript_challenge_bytes",
    "bp_transcript_append_point", "bp_transcript_challenge_scalar", "bp_host_sha3_512", "bp_host_points_sum", "bp_debug_rng_draws", "bp_debug_append_points_x8", "bp_debug_challenge_x8", "bp_r1cs_prove_scenario", "bp_stmt_prover_create", "bp_stmt_free", "bp_stmt_info", "bp_stmt_prove", "bp_stmt_precompute", "bp_stmt_precompute_batch", "bp_gens_share", "bp_r1cs_verification_gh", "bp_r1cs_verify_scenario", "bp_r1cs_batch_verify_scenarios", "bp_ctx_set_profiling", "bp_ctx_kernel_time", "bp_ctx_reset_profiling",
    "bp_debug_decompress", "bp_debug_field_op", "bp_debug_point_op", "bp_debug_glv_decompose", "bp_ctx_set_tuning", "bp_ctx_set_window_shard", "bp_pedersen_commit_batch", "bp_stmt_prover_create_dev",
    # r1cs::ConstraintSystem / Prover / Verifier for the caller's own gadgets
    "bp_prover_new", "bp_verifier_new", "bp_verifier_new_like", "bp_cs_free", "bp_cs_transcript", "bp_cs_metrics", "bp_prover_commit", "bp_verifier_commit",
    "bp_cs_multiply", "bp_cs_allocate", "bp_cs_allocate_multiplier", "bp_cs_constrain", "bp_cs_allocate_multipliers", "bp_cs_constrain_many",
    "bp_cs_specify_randomized_constraints", "bp_cs_challenge_scalar", "bp_prover_set_rng", "bp_prover_precompute", "bp_prover_precompute_batch", "bp_prover_prove",
    "bp_ctx_set_shard_allgather", "bp_gens_fold_tables", "bp_gens_msm_tables", "bp_debug_exp_iter", "bp_debug_inner_product", "bp_verifier_verify", "bp_r1cs_batch_verify", "bp_stmt_as_prover", "bp_transcript_export_state", "bp_transcript_import_state", "bp_transcript_clone",
    "bp_gens_tables_check", "bp_debug_tables_ptr", "bp_rccl_unique_id", "bp_ctx_rccl_init", "bp_ctx_rccl_shutdown", "bp_ctx_collective_stats", "bp_debug_rccl_allgather",
    "bp_debug_vfe_schedule_replay", "bp_debug_vfe_challenges", "bp_ctx_vfe_stats", "bp_debug_verify_challenges", "bp_debug_ctx_create_hostonly", "bp_ctx_msm_stats", "bp_ctx_direct_stats", "bp_gens_direct_tables", "bp_ctx_fold_stats", "bp_gens_fold_tables_slice",
]


class ArkbpError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib().bp_last_error().decode(errors="replace") if _lib is not None else ""
        super().__init__("%s failed with code %d: %s" % (where, code, msg))


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libarkbp_hip.so is not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                "ark_bulletproofs_amd has no CPU fallback"
            )
        _lib = C.CDLL(LIB_PATH)
        _lib.bp_last_error.restype = C.c_char_p
        _lib.bp_transcript_new.restype = C.c_void_p
        _lib.bp_transcript_clone.restype = C.c_void_p
        _lib.bp_cs_transcript.restype = C.c_void_p
        _lib.bp_stmt_as_prover.restype = C.c_void_p
    return _lib


def check(rc, where):
    if rc != BP_OK:
        raise ArkbpError(rc, where)


def ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def u64arr(a, width):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return a.reshape(-1, width)

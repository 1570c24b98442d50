"""ctypes binding of libfsrl_hip.so (C ABI: include/fsrl_hip.h).  Fails loudly if absent."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("FSRL_HIP_LIB") or os.path.join(_HERE, "libfsrl_hip.so")   # override: A/B runs of two builds

FSRL_OK, FSRL_EINVAL, FSRL_ENOMEM, FSRL_EHIP, FSRL_ESTATE = 0, -22, -12, -5, -1
PPO_NSTATS = 11
ALGO_PPO_LAG, ALGO_TRPO_LAG, ALGO_CPO, ALGO_SAC_LAG, ALGO_FOCOPS = 0, 1, 2, 3, 4


class Config(C.Structure):
    """struct fsrl_config (include/fsrl_hip.h)"""
    _fields_ = [("algo", C.c_int32), ("obs_dim", C.c_int32), ("act_dim", C.c_int32),
                ("hidden", C.c_int32), ("n_critics", C.c_int32), ("env_num", C.c_int32),
                ("buffer_size", C.c_int64), ("max_action", C.c_float), ("gamma", C.c_double),
                ("gae_lambda", C.c_double), ("eps_clip", C.c_float), ("dual_clip", C.c_float),
                ("vf_coef", C.c_float), ("max_grad_norm", C.c_float), ("target_kl", C.c_float),
                ("norm_adv", C.c_int32), ("use_lagrangian", C.c_int32), ("lr", C.c_float),
                ("beta1", C.c_float), ("beta2", C.c_float), ("adam_eps", C.c_float), ("recompute_adv", C.c_int32),
                ("unbounded", C.c_int32), ("rew_norm", C.c_int32), ("value_clip", C.c_int32),
                ("hidden1", C.c_int32), ("hidden2", C.c_int32),
                ("n_hidden", C.c_int32), ("hidden_sizes", C.c_int32 * 8), ("force_layered", C.c_int32)]


class ShmEnv(C.Structure):
    """struct fsrl_shm_env (include/fsrl_hip.h): the worker-process env's shared block, as fsrl_collect_run sees it"""
    _fields_ = [("obs", C.c_void_p), ("act", C.c_void_p), ("rew", C.c_void_p), ("cost", C.c_void_p), ("term", C.c_void_p),
                ("trunc", C.c_void_p), ("active", C.c_void_p), ("hs", C.c_void_p), ("want", C.c_void_p), ("owner", C.c_void_p),
                ("lane_of_worker", C.c_void_p), ("env_num", C.c_int32), ("obs_dim", C.c_int32), ("act_dim", C.c_int32),
                ("workers", C.c_int32), ("n_lanes", C.c_int32), ("gen", C.c_uint32 * 2), ("spin", C.c_uint32),
                ("err", C.c_void_p), ("pids", C.c_void_p)]


class TrConfig(C.Structure):
    """struct fsrl_tr_config (include/fsrl_hip.h)"""
    _fields_ = [("target_kl", C.c_float), ("backtrack_coeff", C.c_float), ("damping", C.c_float),
                ("l2_reg", C.c_float), ("critic_lr", C.c_float), ("max_backtracks", C.c_int32),
                ("optim_critic_iters", C.c_int32), ("cg_iters", C.c_int32), ("norm_adv", C.c_int32),
                ("cost_limit", C.c_double)]


class SacConfig(C.Structure):
    """struct fsrl_sac_config (include/fsrl_hip.h)"""
    _fields_ = [("actor_lr", C.c_float), ("critic_lr", C.c_float), ("alpha_lr", C.c_float), ("tau", C.c_float),
                ("alpha", C.c_float), ("target_entropy", C.c_float), ("n_step", C.c_int32),
                ("auto_alpha", C.c_int32), ("use_lagrangian", C.c_int32), ("deterministic", C.c_int32),
                ("exploration_sigma", C.c_float)]


class FocopsConfig(C.Structure):
    """struct fsrl_focops_config (include/fsrl_hip.h)"""
    _fields_ = [("actor_lr", C.c_float), ("critic_lr", C.c_float), ("l2_reg", C.c_float), ("delta", C.c_float),
                ("eta", C.c_float), ("tem_lambda", C.c_float), ("max_grad_norm", C.c_float)]


class CvpoConfig(C.Structure):
    """struct fsrl_cvpo_config (include/fsrl_hip.h)"""
    _fields_ = [("actor_lr", C.c_float), ("critic_lr", C.c_float), ("tau", C.c_float), ("n_step", C.c_int32),
                ("double_critic", C.c_int32), ("sample_act_num", C.c_int32), ("estep_iter_num", C.c_int32),
                ("mstep_iter_num", C.c_int32), ("estep_kl", C.c_float), ("estep_dual_max", C.c_float),
                ("estep_dual_lr", C.c_float), ("mstep_kl_mu", C.c_float), ("mstep_kl_std", C.c_float),
                ("mstep_dual_max", C.c_float), ("mstep_dual_lr", C.c_float), ("qc_thres", C.c_double)]


CPO_NSTATS, TRPO_NSTATS, SAC_NSTATS, FOCOPS_NSTATS, CVPO_NSTATS = 17, 11, 10, 8, 17

_P = C.POINTER
_f, _d, _u8, _i32, _i64 = _P(C.c_float), _P(C.c_double), _P(C.c_uint8), _P(C.c_int32), _P(C.c_int64)
_ctx = C.c_void_p

# name -> (restype, argtypes); every symbol include/fsrl_hip.h declares
SIGNATURES = {
    "fsrl_last_error": (C.c_char_p, []),
    "fsrl_config_default": (None, [_P(Config)]),
    "fsrl_ctx_create": (C.c_int, [C.c_int, _P(Config), _P(_ctx)]),
    "fsrl_ctx_destroy": (C.c_int, [_ctx]),
    "fsrl_sync": (C.c_int, [_ctx]),
    "fsrl_param_count": (C.c_int64, [_ctx]),
    "fsrl_params_set": (C.c_int, [_ctx, _f, C.c_int64]),
    "fsrl_params_get": (C.c_int, [_ctx, _f, C.c_int64]),
    "fsrl_grads_get": (C.c_int, [_ctx, _f, C.c_int64]),
    "fsrl_optim_reset": (C.c_int, [_ctx]),
    "fsrl_state_snapshot": (C.c_int, [_ctx]),
    "fsrl_state_restore": (C.c_int, [_ctx]),
    "fsrl_ret_rms_get": (C.c_int, [_ctx, _d, C.c_int32]),
    "fsrl_ret_rms_set": (C.c_int, [_ctx, _d, C.c_int32]),
    "fsrl_set_lr": (C.c_int, [_ctx, C.c_int32, C.c_float]),
    "fsrl_get_lr": (C.c_float, [_ctx, C.c_int32]),
    "fsrl_ppo_abort": (C.c_int, [_ctx]),
    "fsrl_store_push": (C.c_int, [_ctx, _i32, C.c_int32, _f, _f, _d, _d, _u8, _u8, _f, _i64, _d, _i32, _i64]),
    "fsrl_store_reset": (C.c_int, [_ctx, C.c_int]),
    "fsrl_store_len": (C.c_int64, [_ctx]),
    "fsrl_store_read": (C.c_int, [_ctx, _i64, C.c_int64, _f, _f, _d, _d, _u8, _u8, _f]),
    "fsrl_store_configure": (C.c_int, [_ctx, C.c_int64, C.c_int32]),
    "fsrl_store_geometry": (C.c_int, [_ctx, _i64, _i32]),
    "fsrl_store_sample0": (C.c_int, [_ctx, _i64, C.c_int64, _i64]),
    "fsrl_actor_forward": (C.c_int, [_ctx, _f, C.c_int32, _f, _f]),
    "fsrl_ppo_begin": (C.c_int, [_ctx, _d, C.c_double, C.c_int32, _i64]),
    "fsrl_ppo_pass": (C.c_int, [_ctx, _i64, C.c_uint64, _i32]),
    "fsrl_ppo_pass_result": (C.c_int, [_ctx, _i32]),
    "fsrl_ppo_end": (C.c_int, [_ctx, _f, C.c_int64, _i64]),
    "fsrl_ppo_update": (C.c_int, [_ctx, _d, C.c_double, C.c_int32, C.c_int32, _i64, C.c_uint64, _f,
                                  C.c_int64, _i64, _i32]),
    "fsrl_ppo_set_plan": (C.c_int, [_ctx, C.c_int32]),
    "fsrl_batch_get": (C.c_int, [_ctx, C.c_char_p, _f, C.c_int64]),
    "fsrl_group_create": (C.c_int, [_P(_ctx), C.c_int32, _P(_ctx)]),
    "fsrl_group_destroy": (C.c_int, [_ctx]),
    "fsrl_group_set_plan": (C.c_int, [_ctx, C.c_int32]),
    "fsrl_group_ppo_update": (C.c_int, [_ctx, _d, _d, C.c_int32, C.c_int32, _P(_i64), C.c_uint64, _P(_f), C.c_int64, _i64,
                                        _i32]),
    "fsrl_gae_return": (C.c_int, [_ctx, _f, _f, _d, _u8, C.c_int64, C.c_double, C.c_double, _d]),
    "fsrl_nstep_return": (C.c_int, [_ctx, _d, _u8, C.c_int64, _f, _i64, C.c_int64, C.c_int64, C.c_double, C.c_int32, _d]),
    "fsrl_launch_floors": (C.c_int, [_ctx, C.c_int32, C.c_int32, _d]),
    "fsrl_tr_begin": (C.c_int, [_ctx, _P(TrConfig), _i64]),
    "fsrl_cpo_learn": (C.c_int, [_ctx, C.c_double, C.c_int32, _f]),
    "fsrl_trpo_learn": (C.c_int, [_ctx, _d, C.c_double, C.c_int32, _f]),
    "fsrl_cpo_learn_mb": (C.c_int, [_ctx, C.c_double, C.c_int32, C.c_int32, _i64, C.c_uint64, _f, C.c_int64, _i64]),
    "fsrl_trpo_learn_mb": (C.c_int, [_ctx, _d, C.c_double, C.c_int32, C.c_int32, _i64, C.c_uint64, _f, C.c_int64, _i64]),
    "fsrl_tr_linesearch_evals": (C.c_int32, [_ctx, _i32, C.c_int32]),
    "fsrl_actor_param_count": (C.c_int64, [_ctx]),
    "fsrl_tr_grad": (C.c_int, [_ctx, C.c_int32, _f, C.c_int64]),
    "fsrl_tr_hvp": (C.c_int, [_ctx, _f, _f, C.c_int64]),
    "fsrl_tr_hvp_cached": (C.c_int, [_ctx, _f, _f, C.c_int64]),
    "fsrl_tr_eval": (C.c_int, [_ctx, _d]),
    "fsrl_tr_set_plan": (C.c_int, [_ctx, C.c_int32, C.c_int32, C.c_int32]),
    "fsrl_tr_set_tile_split": (C.c_int, [_ctx, C.c_int32, C.c_int32]),
    "fsrl_tr_set_co_delay": (C.c_int, [_ctx, C.c_int32, C.c_int32]),
    "fsrl_focops_init": (C.c_int, [_ctx, _P(FocopsConfig)]),
    "fsrl_focops_set_plan": (C.c_int, [_ctx, C.c_int32]),
    "fsrl_focops_set_nu": (C.c_int, [_ctx, C.c_double, C.c_double]),
    "fsrl_sac_init": (C.c_int, [_ctx, _P(SacConfig)]),
    "fsrl_sac_set_plan": (C.c_int, [_ctx, C.c_int32]),
    "fsrl_sac_param_count": (C.c_int64, [_ctx, C.c_int32]),
    "fsrl_sac_params_set": (C.c_int, [_ctx, _f, C.c_int64, _f, C.c_int64, C.c_float]),
    "fsrl_sac_params_put": (C.c_int, [_ctx, C.c_int32, _f, C.c_int64]),
    "fsrl_sac_params_get": (C.c_int, [_ctx, C.c_int32, _f, C.c_int64, _f]),
    "fsrl_sac_update": (C.c_int, [_ctx, C.c_int32, _i64, _f, _f, C.c_uint64, _d, C.c_double, _f]),
    "fsrl_actor_sample": (C.c_int, [_ctx, _f, C.c_int32, C.c_int32, C.c_uint64, _f]),
    "fsrl_collect_step": (C.c_int, [_ctx, _i32, C.c_int32, _f, _f, _d, _d, _u8, _u8, _f, _i64, _d, _i32, _i64,
                                    _f, C.c_int32, C.c_int32, C.c_int32, _f, _f, _f, _f]),
    "fsrl_collect_episodes": (C.c_int, [_ctx, _P(ShmEnv), _i32, C.c_int32, _f, C.c_int32, C.c_int32, C.c_int32, _f, _f, _i64, _d, _i32,
                                        _i32, _d, _i32, _i32]),
    "fsrl_collect_episodes_split": (C.c_int, [_ctx, _P(ShmEnv), _i32, C.c_int32, _f, C.c_int32, C.c_int32, C.c_int32, _f, _f, _i64, _d, _i32,
                                              _i32, _d, _i32, _i32]),
    "fsrl_collect_timing": (C.c_int, [_ctx, _d]),
    "fsrl_actor_set_resident": (C.c_int, [_ctx, C.c_int32, C.c_double]),
    "fsrl_actor_resident_stats": (C.c_int, [_ctx, _i64]),
    "fsrl_actor_release": (C.c_int, [_ctx]),
    "fsrl_collect_run": (C.c_int, [_ctx, _P(ShmEnv), _i32, C.c_int32, _f, _f, _f, C.c_int32, C.c_int32, _f, _f, C.c_int32, _i32, _d, _d, _d,
                                  _u8, _u8, _f]),
    "fsrl_store_sizes": (C.c_int, [_ctx, _i64, C.c_int32]),
    "fsrl_sac_stats_drain": (C.c_int64, [_ctx, _f, C.c_int64]),
    "fsrl_sac_last_sample": (C.c_int, [_ctx, _i64, _f, _f, C.c_int32]),
    "fsrl_sac_actor_forward": (C.c_int, [_ctx, _f, C.c_int32, _f, _f]),
    "fsrl_cvpo_init": (C.c_int, [_ctx, _P(CvpoConfig)]),
    "fsrl_cvpo_pre_update": (C.c_int, [_ctx]),
    "fsrl_cvpo_post_update": (C.c_int, [_ctx]),
    "fsrl_cvpo_set_thres": (C.c_int, [_ctx, C.c_double]),
    "fsrl_cvpo_update": (C.c_int, [_ctx, C.c_int32, _i64, _f, _f, C.c_uint64, _f]),
    "fsrl_cvpo_duals_get": (C.c_int, [_ctx, _f]),
    "fsrl_cvpo_last_particles": (C.c_int, [_ctx, _f, C.c_int64]),
    "fsrl_comm_unique_id": (C.c_int, [_u8, C.c_int32]),
    "fsrl_comm_init": (C.c_int, [_ctx, C.c_int32, C.c_int32, _u8, C.c_int32]),
    "fsrl_metrics_allreduce": (C.c_int, [_ctx, _d, C.c_int32]),
    "fsrl_comm_info": (C.c_int, [_ctx, _i32, _i32]),
    "fsrl_comm_destroy": (C.c_int, [_ctx]),
    "fsrl_set_profiling": (C.c_int, [_ctx, C.c_int]),
    "fsrl_last_timing": (C.c_int, [_ctx, _d, C.c_int32]),
}

_lib = None


def load():
    """Load the HIP library; raise (never fall back) if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with fsrl_amd/csrc/build.sh (hipcc, gfx950). "
            "fsrl_amd has no CPU fallback.")
    # PyTorch-ROCm ships its own HIP runtime; whichever copy initialises first owns the process.  Loading
    # ours first leaves torch.cuda (RCCL, the bench's barriers) -- or, the other way round, this library --
    # without a device ("no ROCm-capable device is detected").  torch first is the order that works.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the ABI and the header drifted apart
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class FsrlHipError(RuntimeError):
    pass


def check(rc):
    """Map C status codes to the exception types the reference raises for the same fault
    (assert -> AssertionError for argument violations; RuntimeError otherwise)."""
    if rc == FSRL_OK:
        return
    msg = load().fsrl_last_error().decode("utf-8", "replace")
    if rc == FSRL_EINVAL:
        raise AssertionError(msg)
    if rc == FSRL_ENOMEM:
        raise MemoryError(msg)
    raise FsrlHipError(f"[{rc}] {msg}")

"""ctypes binding of libses_hip.so (C ABI: include/ses.h).

The library is the product: there is NO CPU fallback.  If it is missing or no MI355X is visible,
everything here raises -- loudly -- instead of computing anything on the host.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SES_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "libses_hip.so")  # env: dev A/B builds

SES_OK = 0
ENV_NONE = -1
ENV_CARTPOLE = 0
ENV_LUNARLANDER = 1
ENV_SIMPLE_SPREAD = 2
ENV_BIPEDALWALKER = 3
MODE_EPISODIC = 0
MODE_FIXED_LENGTH = 1
HIDDEN = 32


class SesConfig(ctypes.Structure):
    _fields_ = [
        ("env_id", ctypes.c_int32),
        ("num_state", ctypes.c_int32),
        ("num_action", ctypes.c_int32),
        ("discrete_action", ctypes.c_int32),
        ("gru", ctypes.c_int32),
        ("pomdp", ctypes.c_int32),
        ("max_step", ctypes.c_int32),
        ("eval_ep_num", ctypes.c_int32),
        ("device", ctypes.c_int32),
        ("lanes_per_env", ctypes.c_int32),
        ("n_agents", ctypes.c_int32),
        ("physics64", ctypes.c_int32),
    ]


class SesGenState(ctypes.Structure):
    """ses_gen_state of include/ses.h (ses_run_generations)."""
    _fields_ = [
        ("strategy", ctypes.c_int32), ("n", ctypes.c_int32), ("elite_num", ctypes.c_int32), ("mode", ctypes.c_int32),
        ("shared_init", ctypes.c_int32), ("init_width", ctypes.c_int32),
        ("init_lo", ctypes.c_float), ("init_hi", ctypes.c_float),
        ("seed", ctypes.c_uint64), ("env_seed", ctypes.c_uint64),
        ("learning_rate", ctypes.c_double), ("sigma_decay", ctypes.c_double),
        ("sigma", ctypes.c_double), ("pop_sigma", ctypes.c_double),
        ("pop_gen", ctypes.c_uint64), ("adam_t", ctypes.c_int64),
        ("cur", ctypes.c_int32), ("world", ctypes.c_int32),
        ("theta", ctypes.c_void_p * 2), ("parents", ctypes.c_void_p * 2),
        ("adam_m", ctypes.c_void_p * 2), ("adam_v", ctypes.c_void_p * 2),
        ("parent_map", ctypes.c_void_p), ("alias_state", ctypes.c_void_p),
        ("fitness", ctypes.c_void_p), ("init", ctypes.c_void_p),
        ("work_i32", ctypes.c_void_p), ("work_f32", ctypes.c_void_p),
        ("first_row", ctypes.c_int64), ("n_local", ctypes.c_int32), ("per_rank", ctypes.c_int32),
        ("comm", ctypes.c_void_p), ("fit_local", ctypes.c_void_p),
    ]


STRATEGY_OPENAI_ES, STRATEGY_SIMPLE_EVOLUTION, STRATEGY_SIMPLE_GENETIC = 0, 1, 2

_vp = ctypes.c_void_p
_i32 = ctypes.c_int32
_i64 = ctypes.c_int64
_u64 = ctypes.c_uint64
_f32 = ctypes.c_float
_f64 = ctypes.c_double

# name -> argtypes; every entry point declared in include/ses.h (tests/test_abi.py checks the two agree)
SIGNATURES = {
    "ses_create": [ctypes.POINTER(SesConfig), _vp, ctypes.POINTER(_vp)],
    "ses_destroy": [_vp],
    "ses_sync": [_vp],
    "ses_set_tuning": [_vp, ctypes.c_char_p, _i32],
    "ses_set_stamp": [_vp, _vp],
    "ses_last_error": [],
    "ses_version": [],
    "ses_param_count": [_i32, _i32, _i32],
    "ses_device_count": [],
    "ses_perturb": [_vp, _vp, _vp, _vp, _f32, _u64, _u64, _i64, _i32, _vp],
    "ses_noise": [_vp, _u64, _u64, _i64, _i32, _vp],
    "ses_perturb_host_noise": [_vp, _vp, _vp, _vp, _f64, _i32, _vp, _vp],
    "ses_init_states_uniform": [_vp, _u64, _u64, _i64, _i32, _i32, _i32, _f32, _f32, _vp],
    "ses_init_states_uniform_gens": [_vp, _u64, _u64, _i32, _i64, _i32, _i32, _i32, _f32, _f32, _vp],
    "ses_policy_forward": [_vp, _vp, _vp, _vp, _i32, _vp, _vp, _vp],
    "ses_env_step": [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ses_env_state_bytes": [_vp],
    "ses_env_obs_width": [_vp],
    "ses_env_reset": [_vp, _vp, _i32, _vp, _vp],
    "ses_env_step_generic": [_vp, _vp, _vp, _i32, _vp, _vp, _vp],
    "ses_env_step_shape": [_vp, _vp, _vp, _vp, _vp],
    "ses_stream_probe": [_vp, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "ses_rollout": [_vp, _vp, _vp, _i32, _i32, _i32, _vp, _vp, _vp],
    "ses_rank_center": [_vp, _vp, _i32, _vp, _vp, _vp],
    "ses_es_update_philox": [_vp, _vp, _i32, _i32, _u64, _u64, _f64, _f64, _f64, _vp, _vp, _vp, _vp],
    "ses_openai_generation": [_vp, _vp, _i32, _u64, _u64, _f64, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _u64, _i64, _i32,
                              _vp, _vp],
    "ses_openai_sharded_ok": [_vp, _vp, _i32, _i32, _i32],
    "ses_openai_generation_sharded": [_vp, _vp, _vp, _i32, _u64, _u64, _f64, _f64, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _f32, _u64,
                                      _i64, _i32, _i32, _i32, _vp, _vp],
    "ses_es_update_stored": [_vp, _vp, _i32, _vp, _f64, _f64, _f64, _vp, _vp, _vp, _vp],
    "ses_elite_ids": [_vp, _vp, _i32, _i32, _vp],
    "ses_elite_select": [_vp, _vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp],
    "ses_elite_mean": [_vp, _vp, _vp, _i32, _vp],
    "ses_gather_rows": [_vp, _vp, _vp, _i32, _vp],
    "ses_run_generations": [_vp, ctypes.POINTER(SesGenState), _i32, _vp, _vp],
    "ses_comm_unique_id": [_vp],
    "ses_comm_init": [_vp, _i32, _i32, _vp],
    "ses_comm_info": [_vp, _vp, _vp, _vp],
    "ses_comm_destroy": [_vp],
    "ses_allgather_fitness": [_vp, _vp, _i32, _vp],
    "ses_comm_p2p_export": [_vp, _i32, _i32, _i32, _vp],
    "ses_comm_p2p_attach": [_vp, _vp],
    "ses_comm_p2p_attach_local": [_vp, _vp],
    "ses_stream_create_exclusive": [_i32, _vp],
    "ses_stream_destroy": [_vp],
    "ses_comm_p2p_info": [_vp, _vp, _vp, _vp],
    "ses_comm_p2p_counts": [_vp, _vp, _vp],
    "ses_comm_p2p_status": [_vp, _vp],
    "ses_comm_p2p_reset_status": [_vp],
    "ses_comm_p2p_detach": [_vp],
}
COMM_ID_BYTES = 128
COMM_P2P_HANDLE_BYTES = 64
_RESTYPE = {"ses_last_error": ctypes.c_char_p, "ses_version": ctypes.c_char_p}

_lib = None


class SesError(RuntimeError):
    pass


def load():
    """dlopen libses_hip.so; raises if it has not been built (python __graft_entry__.py / csrc/build.sh)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SesError(f"{LIB_PATH} not found: build it with simple-es_amd/csrc/build.sh "
                       f"(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError here == ABI drift; let it surface
        fn.argtypes = argtypes
        fn.restype = _RESTYPE.get(name, ctypes.c_int)
    _lib = lib
    return lib


def check(rc, what=""):
    if rc != SES_OK:
        msg = load().ses_last_error().decode("utf-8", "replace")
        raise SesError(f"{what or 'libses_hip'} failed (code {rc}): {msg}")
    return rc

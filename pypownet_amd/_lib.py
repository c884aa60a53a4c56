"""ctypes binding of libppn.so (include/ppn.h).  The HIP library is mandatory: there is NO CPU fallback --
importing the engine without the built extension raises, loudly."""
import ctypes as C
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libppn.so')


class PpnCase(C.Structure):
    _fields_ = [('n_bus_rows', C.c_int32), ('bus_cols', C.c_int32), ('n_gen', C.c_int32), ('gen_cols', C.c_int32),
                ('n_branch', C.c_int32), ('branch_cols', C.c_int32), ('base_mva', C.c_double),
                ('bus', C.POINTER(C.c_double)), ('gen', C.POINTER(C.c_double)), ('branch', C.POINTER(C.c_double))]


REWARD_PARAM_NAMES = ['line_usage', 'distance_initial_grid', 'number_loads_cut', 'number_prods_cut', 'loadflow_exception',
                      'illegal_broken_line_switch', 'illegal_oncooldown_line_switch',
                      'illegal_oncooldown_substation_switch', 'too_many_productions_cut', 'too_many_consumptions_cut',
                      'too_much_activated_elements', 'number_line_switches', 'number_node_switches']


class PpnRewardParams(C.Structure):
    """ppn_reward_params (include/ppn.h): coefficients of the reference's five-component reward."""
    _fields_ = [(k, C.c_double) for k in REWARD_PARAM_NAMES]


class PpnRules(C.Structure):
    _fields_ = [('mode', C.c_int32), ('solver', C.c_int32), ('tol', C.c_double), ('max_it', C.c_int32),
                ('hard_overflow_coefficient', C.c_double),
                ('n_timesteps_hard_overflow_is_broken', C.c_int32),
                ('n_timesteps_consecutive_soft_overflow_breaks', C.c_double),
                ('n_timesteps_soft_overflow_is_broken', C.c_int32),
                ('n_timesteps_horizon_maintenance', C.c_int32),
                ('max_number_prods_game_over', C.c_int32), ('max_number_loads_game_over', C.c_int32),
                ('n_timesteps_actionned_line_reactionable', C.c_int32),
                ('n_timesteps_actionned_node_reactionable', C.c_int32),
                ('max_number_actionned_substations', C.c_int32), ('max_number_actionned_lines', C.c_int32),
                ('max_number_actionned_total', C.c_int32), ('game_over_mode_hard', C.c_int32),
                ('chronic_looping', C.c_int32), ('max_active_buses', C.c_int32), ('lu_capacity', C.c_int32),
                ('rng_seed', C.c_int32), ('q_plane_auto', C.c_int32)]


class PpnMpcBatch(C.Structure):
    """ppn_mpc_batch (include/ppn.h): MATPOWER arrays of n cases in, result arrays out."""
    _dp = C.POINTER(C.c_double)
    _fields_ = [('struct_size', C.c_int32), ('n', C.c_int32), ('bus_cols', C.c_int32), ('gen_cols', C.c_int32), ('branch_cols', C.c_int32),
                ('bus_rows', C.c_int32), ('gen_rows', C.c_int32), ('branch_rows', C.c_int32),
                ('bus', _dp), ('gen', _dp), ('branch', _dp), ('bus_out', _dp), ('gen_out', _dp), ('branch_out', _dp),
                ('success', C.POINTER(C.c_uint8)), ('outcome', C.POINTER(C.c_int32))]


class PpnAsyncConfig(C.Structure):
    """ppn_async_config (include/ppn.h): the asynchronous session's observation layout, server size and the caller's device buffers."""
    _fields_ = [('struct_size', C.c_int32), ('layout', C.c_int32), ('as_f32', C.c_int32), ('workgroups', C.c_int32),
                ('idle_timeout_ms', C.c_int32), ('reserved', C.c_int32), ('obs_device', C.c_void_p), ('obs_bytes', C.c_size_t),
                ('report_device', C.c_void_p)]


class PpnChronic(C.Structure):
    _fields_ = [('T', C.c_int32)] + [(k, C.POINTER(C.c_float)) for k in (
        'prods_p', 'prods_v', 'loads_p', 'loads_q', 'prods_p_planned', 'prods_v_planned', 'loads_p_planned',
        'loads_q_planned', 'maintenance', 'hazards')] + [('ids', C.POINTER(C.c_int32)),
                                                          ('dates', C.POINTER(C.c_int32))]


# ppn_field enum (include/ppn.h)
FIELDS = ['VM', 'VA', 'PG', 'QG', 'VG', 'PD', 'QD', 'PF', 'QF', 'PT', 'QT', 'AMPS', 'PRODS_NODES', 'LOADS_NODES',
          'LINES_OR_NODES', 'LINES_EX_NODES', 'LINES_STATUS', 'RECONNECTABLE', 'LINE_COOLDOWN', 'NODE_COOLDOWN',
          'SOFT_COUNT', 'DONE', 'FLAG', 'ILLEGAL', 'CASCADE_DEPTH', 'N_SOLVES', 'N_ITERS', 'CHRONIC_SLOT',
          'CHRONIC_ROW', 'N_LOADS_CUT', 'N_PRODS_CUT', 'SUCCESS', 'OBSERVATION', 'BUS_TYPE', 'REWARD', 'ILLEGAL_COUNTS',
          'ACTION_SWITCHES', 'LINE_EVENTS', 'SOLVE_OUTCOME', 'N_STEPS', 'RETURN', 'DEAD', 'EPOCH', 'STEP_REPORT']
FIELD_ID = {k: i for i, k in enumerate(FIELDS)}
_F64 = {'VM', 'VA', 'PG', 'QG', 'VG', 'PD', 'QD', 'PF', 'QF', 'PT', 'QT', 'AMPS', 'OBSERVATION', 'REWARD', 'RETURN', 'STEP_REPORT'}
_U8 = {'PRODS_NODES', 'LOADS_NODES', 'LINES_OR_NODES', 'LINES_EX_NODES', 'LINES_STATUS', 'DONE', 'SUCCESS', 'BUS_TYPE',
       'LINE_EVENTS', 'DEAD'}


def field_dtype(name):
    import numpy as np
    if name in _F64:
        return np.float64
    if name in _U8:
        return np.uint8
    return np.int32


EXPORTS = ['ppn_create', 'ppn_destroy', 'ppn_last_error', 'ppn_set_thermal_limits', 'ppn_load_chronic', 'ppn_reset',
           'ppn_step', 'ppn_process_game_over', 'ppn_is_action_valid', 'ppn_runpf_batch', 'ppn_field_bytes',
           'ppn_read', 'ppn_write', 'ppn_sync', 'ppn_stream', 'ppn_kernel_time', 'ppn_dim', 'ppn_version', 'ppn_set_reward',
           'ppn_simulate_candidates', 'ppn_read_observation', 'ppn_observation_length', 'ppn_wait', 'ppn_runpf_arrays', 'ppn_rollout',
           'ppn_policy_actions', 'ppn_rollout_policy', 'ppn_step_observe', 'ppn_async_start', 'ppn_send', 'ppn_recv', 'ppn_async_stop',
           'ppn_async_stream', 'ppn_async_stat', 'ppn_restart_memo', 'ppn_restart_memo_stat']


def _preload_torch_hip_runtime():
    import importlib.util
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec is None or not spec.submodule_search_locations:
        return
    if os.environ.get('PPN_IMPORT_TORCH') == '1':
        import torch  # noqa: F401
        return
    path = os.path.join(list(spec.submodule_search_locations)[0], 'lib', 'libamdhip64.so')
    if os.path.exists(path):
        try:
            C.CDLL(path, mode=C.RTLD_GLOBAL)
            return
        except OSError:
            pass
    try:
        import torch  # noqa: F401
    except ImportError:
        pass


def load_library():
    """Loads pypownet_amd/libppn.so -- the one and only library of the product path -- and declares its signatures."""
    # PyTorch-ROCm ships its own copy of the HIP runtime.  One process must not end up with two: whichever libamdhip64 is
    # loaded first serves both, and a torch that initialises its device AFTER libppn.so brought in the system runtime finds
    # "No HIP GPUs".  So when torch is installed but not imported yet, ITS runtime library is loaded first -- the shared object
    # alone, not the framework (importing torch costs seconds and pulls a whole framework into plain RunEnv users);
    # PPN_IMPORT_TORCH=1 falls back to importing torch itself.
    if 'torch' not in sys.modules:
        _preload_torch_hip_runtime()
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            'pypownet_amd: the HIP extension %s is missing. Build it with `python __graft_entry__.py` '
            '(hipcc --offload-arch=gfx950); this engine has no CPU fallback.' % LIB_PATH)
    return bind_signatures(C.CDLL(LIB_PATH))


def bind_signatures(lib, full_abi=True):
    """Declares argtypes / restypes of include/ppn.h on a loaded library object (full_abi=False: the subset a checker
    library of the test-suite exports)."""
    vp = C.c_void_p
    lib.ppn_create.argtypes = [C.POINTER(PpnCase), C.POINTER(PpnRules), C.c_int32, C.c_int32, C.POINTER(vp)]
    lib.ppn_create.restype = C.c_int
    lib.ppn_destroy.argtypes = [vp]
    lib.ppn_destroy.restype = C.c_int
    lib.ppn_last_error.argtypes = [vp]
    lib.ppn_last_error.restype = C.c_char_p
    if full_abi:
        lib.ppn_set_reward.argtypes = [vp, C.POINTER(PpnRewardParams)]
        lib.ppn_set_reward.restype = C.c_int
        lib.ppn_simulate_candidates.argtypes = [vp, vp, C.c_int32, C.POINTER(C.c_int32), C.c_int32]
        lib.ppn_simulate_candidates.restype = C.c_int
        lib.ppn_read_observation.argtypes = [vp, C.c_int32, C.c_int32, vp, C.c_size_t, C.c_int32, C.c_int32]
        lib.ppn_read_observation.restype = C.c_int
        lib.ppn_observation_length.argtypes = [vp, C.c_int32]
        lib.ppn_observation_length.restype = C.c_int32
        lib.ppn_rollout.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32]
        lib.ppn_rollout.restype = C.c_int
        lib.ppn_runpf_arrays.argtypes = [vp, C.POINTER(PpnMpcBatch)]
        lib.ppn_runpf_arrays.restype = C.c_int
    lib.ppn_set_thermal_limits.argtypes = [vp, C.POINTER(C.c_double)]
    lib.ppn_set_thermal_limits.restype = C.c_int
    lib.ppn_load_chronic.argtypes = [vp, C.c_int32, C.POINTER(PpnChronic)]
    lib.ppn_load_chronic.restype = C.c_int
    lib.ppn_reset.argtypes = [vp, C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    lib.ppn_reset.restype = C.c_int
    lib.ppn_step.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32]
    lib.ppn_step.restype = C.c_int
    lib.ppn_process_game_over.argtypes = [vp, vp]
    lib.ppn_process_game_over.restype = C.c_int
    lib.ppn_is_action_valid.argtypes = [vp, vp, vp]
    lib.ppn_is_action_valid.restype = C.c_int
    lib.ppn_runpf_batch.argtypes = [vp]
    lib.ppn_runpf_batch.restype = C.c_int
    lib.ppn_field_bytes.argtypes = [vp, C.c_int]
    lib.ppn_field_bytes.restype = C.c_size_t
    lib.ppn_read.argtypes = [vp, C.c_int, vp, C.c_size_t, C.c_int32, C.c_int32]
    lib.ppn_read.restype = C.c_int
    lib.ppn_write.argtypes = [vp, C.c_int, vp, C.c_size_t]
    lib.ppn_write.restype = C.c_int
    lib.ppn_sync.argtypes = [vp]
    lib.ppn_sync.restype = C.c_int
    if full_abi:
        lib.ppn_wait.argtypes = [vp]
        lib.ppn_wait.restype = C.c_int
        lib.ppn_policy_actions.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.c_int32, vp]
        lib.ppn_policy_actions.restype = C.c_int
        lib.ppn_rollout_policy.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.c_int32, C.c_int32]
        lib.ppn_rollout_policy.restype = C.c_int
        lib.ppn_step_observe.argtypes = [vp, vp, C.c_int32, C.c_int32, C.c_int32, C.c_int32, vp, C.c_size_t]
        lib.ppn_step_observe.restype = C.c_int
        lib.ppn_async_start.argtypes = [vp, C.POINTER(PpnAsyncConfig)]
        lib.ppn_async_start.restype = C.c_int
        lib.ppn_send.argtypes = [vp, vp, C.c_int32, vp, C.c_int32, C.c_int32]
        lib.ppn_send.restype = C.c_int
        lib.ppn_recv.argtypes = [vp, C.c_int32, C.c_int32, C.c_int32, vp, C.POINTER(C.c_int32), vp]
        lib.ppn_recv.restype = C.c_int
        lib.ppn_async_stop.argtypes = [vp]
        lib.ppn_async_stop.restype = C.c_int
        lib.ppn_async_stream.argtypes = [vp]
        lib.ppn_async_stream.restype = vp
        lib.ppn_async_stat.argtypes = [vp, C.c_int32]
        lib.ppn_async_stat.restype = C.c_int64
        lib.ppn_restart_memo.argtypes = [vp, C.c_int32, C.c_int64]
        lib.ppn_restart_memo.restype = C.c_int
        lib.ppn_restart_memo_stat.argtypes = [vp, C.c_int32]
        lib.ppn_restart_memo_stat.restype = C.c_int64
    lib.ppn_stream.argtypes = [vp]
    lib.ppn_stream.restype = vp
    lib.ppn_kernel_time.argtypes = [vp, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_int64)]
    lib.ppn_kernel_time.restype = C.c_int
    lib.ppn_dim.argtypes = [vp, C.c_int32]
    lib.ppn_dim.restype = C.c_int32
    lib.ppn_version.argtypes = []
    lib.ppn_version.restype = C.c_char_p
    return lib

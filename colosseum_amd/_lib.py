"""ctypes binding of libcmdp.so (include/cmdp.h).  Fails loudly when the HIP library is missing:
there is no CPU fallback anywhere in this package."""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "libcmdp.so")

OK, ERR_INVALID, ERR_HIP, ERR_NEEDS_RESET, ERR_UNSUPPORTED, ERR_MAX_ITER, ERR_NO_DEVICE, ERR_MAX_VALUE, ERR_OVERFLOW = (
    0, -1, -2, -3, -4, -5, -6, -7, -8)
RNG_MT_COMPAT, RNG_PHILOX = 0, 1
POLICY_RANDOM, POLICY_HOST_ACTIONS, POLICY_GREEDY_Q = 0, 1, 2
SCHEME_AUTO, SCHEME_JACOBI, SCHEME_GAUSS_SEIDEL = 0, 1, 2
LAYOUT_CSR, LAYOUT_DENSE = 0, 1
FLAG_REWARD_MEANS = 1
FLAG_BETA_GAMMAS = 4   # device-sampled Beta rewards: two gammas for every shape (rounds 1-2 recipe)
FLAG_REWARD_CACHE = 2  # reference-exact per-triple reward caches for batches (csrc/cmdp_reward_cache.h)
OPT_ROLLOUT_KERNEL = 1
OPT_DP_KERNEL = 2
OPT_LDS_GROUPS_PER_CU = 3
OPT_DIAMETER_WORKSPACE_MB = 4
OPT_CHAIN_EXACT_ORDER = 5
OPT_MIXING_PATH = 6
OPT_DIAMETER_RELABEL_MIN_STATES = 7
STAT_DP_KERNEL_MS, STAT_DP_KERNEL, STAT_REWARD_FILLS, STAT_REWARD_ROUNDS, STAT_ROLLOUT_KERNEL_MS, STAT_HIST_KERNEL_MS = 1, 2, 3, 4, 5, 6
STAT_CHAIN_FAST_INSTANCES = 7
STAT_REWARD_FILL_MS, STAT_REWARD_ROUND_MS = 8, 9
STAT_DIAMETER_CLUSTER_LAUNCHES, STAT_DIAMETER_CLUSTER_FALLBACKS = 10, 11
NOISE_NONE, NOISE_GAUSSIAN, NOISE_GAUSSIAN_CORRELATED, NOISE_STUDENT_T, NOISE_STUDENT_T_CORRELATED = 0, 1, 2, 3, 4
CALIB_LDS_READ, CALIB_LDS_CHAIN, CALIB_LDS_CHAIN_SHARED = 0, 1, 2
DP_AUTO, DP_WORKGROUP, DP_REGISTER = 0, 1, 2
DP_REGISTER_DISTINCT = 5  # K2U: register-resident, gathers deduplicated per state
DP_REGISTER_WAVEFRONT = 7  # K2W: K2U's tables with one wavefront per instance (<= 448 states, <= 5 distinct successors)
ROLLOUT_AUTO, ROLLOUT_GLOBAL, ROLLOUT_LDS, ROLLOUT_LDS_STOCHASTIC = 0, 1, 2, 3
ROLLOUT_LDS_TEMPLATE = 4  # K1T: K1P with one successor table per workgroup (batches of action-permuted copies of one MDP)
ROLLOUT_LDS_TEMPLATE_STREAM = 5  # K1U: K1T with the trace streamed to HBM and histogrammed afterwards (256 instances per CU)
ROLLOUT_EPISODE_PARALLEL = 6  # K1E: lane = (instance, episode): episodic batches with two actions walk their episodes in parallel

EXPORTS = [
    "cmdp_version", "cmdp_build_id", "cmdp_last_error", "cmdp_device_count", "cmdp_set_device", "cmdp_create", "cmdp_destroy",
    "cmdp_stream", "cmdp_reset", "cmdp_step", "cmdp_rollout", "cmdp_rollout_async", "cmdp_synchronize", "cmdp_stat", "cmdp_calibrate", "cmdp_set_option", "cmdp_lds_plan",
    "cmdp_visits", "cmdp_reset_visits", "cmdp_set_visits", "cmdp_state", "cmdp_last_start", "cmdp_vi_discounted", "cmdp_pe_discounted",
    "cmdp_vi_episodic", "cmdp_pe_episodic", "cmdp_diameter", "cmdp_diameter_episodic", "cmdp_value_norm", "cmdp_gth", "cmdp_qlearning_create", "cmdp_qlearning_destroy", "cmdp_qlearning_run",
    "cmdp_qlearning_tables", "cmdp_qlearning_evaluate", "cmdp_greedy_policy_episodic", "cmdp_qlearning_continuous_create",
    "cmdp_qlearning_policy", "cmdp_qlearning_average_reward", "cmdp_qlearning_run_logged", "cmdp_tracker_replay", "cmdp_average_reward", "cmdp_diameter_range", "cmdp_diameter_sparse_f64", "cmdp_mixing_time", "cmdp_set_observation_table", "cmdp_observe", "cmdp_observe_noise",
    "cmdp_set_reward_streams", "cmdp_legacy_beta",
]


class CmdpDesc(C.Structure):
    _fields_ = [
        ("n_instances", C.c_int32), ("n_actions", C.c_int32), ("horizon", C.c_int32), ("rng_mode", C.c_int32),
        ("layout", C.c_int32), ("flags", C.c_int32),
        ("reward_min", C.c_double), ("reward_max", C.c_double),
        ("state_off", C.c_void_p),
        ("sp_ptr", C.c_void_p), ("sp_next", C.c_void_p), ("sp_cum", C.c_void_p), ("sp_reward", C.c_void_p),
        ("sp_rkind", C.c_void_p), ("sp_rp0", C.c_void_p), ("sp_rp1", C.c_void_p), ("sp_seed", C.c_void_p), ("start_off", C.c_void_p), ("start_state", C.c_void_p),
        ("start_cum", C.c_void_p), ("start_seed", C.c_void_p), ("philox_key", C.c_void_p),
        ("csr_ptr", C.c_void_p), ("csr_col", C.c_void_p), ("csr_val", C.c_void_p), ("R", C.c_void_p),
    ]


class CmdpLoopDesc(C.Structure):
    _fields_ = [
        ("n_steps", C.c_int64), ("log_every", C.c_int64),
        ("n_check", C.c_int32), ("horizon", C.c_int32), ("kmax", C.c_int32), ("reserved", C.c_int32),
        ("max_time", C.c_double),
        ("base_val", C.c_void_p), ("base_kind", C.c_void_p), ("opt0", C.c_void_p), ("worst0", C.c_void_p),
        ("start_pos", C.c_void_p), ("start_prob", C.c_void_p),
    ]


LOG_COLUMNS = [  # CMDP_LOG_COLUMNS: the indicator names in sorted order, without "steps"
    "cumulative_expected_reward", "cumulative_regret", "cumulative_reward", "normalized_cumulative_expected_reward",
    "normalized_cumulative_regret", "normalized_cumulative_reward", "optimal_cumulative_expected_reward",
    "optimal_normalized_cumulative_expected_reward", "random_cumulative_expected_reward", "random_cumulative_regret",
    "random_normalized_cumulative_expected_reward", "random_normalized_cumulative_regret", "steps_per_second",
    "worst_cumulative_expected_reward", "worst_cumulative_regret", "worst_normalized_cumulative_expected_reward",
    "worst_normalized_cumulative_regret",
]


class CmdpError(RuntimeError):
    def __init__(self, code, message):
        super().__init__(f"libcmdp error {code}: {message}")
        self.code = code


class DynamicProgrammingMaxIterationExceeded(Exception):
    """Same name as the reference's exception (colosseum/dynamic_programming/utils.py:8-9)."""


_lib = None


def source_files():
    """The files libcmdp.so is compiled from, in the order their hash is taken."""
    csrc = os.path.join(HERE, "csrc")
    files = [os.path.join(csrc, f) for f in sorted(os.listdir(csrc)) if f.endswith((".hip", ".h"))]
    files.append(os.path.join(os.path.dirname(HERE), "include", "cmdp.h"))
    return files


def source_hash():
    """SHA-256 over the library's sources: what `cmdp_build_id()` of an up-to-date build returns."""
    import hashlib

    h = hashlib.sha256()
    for f in source_files():
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()


def built_id(path=LIB_PATH):
    """The build id stamped into the library file at `path`, read from the file's bytes (no dlopen: a process that
    rebuilds the library must not have the old image mapped under the same name); None if missing or unstamped."""
    import re

    if not os.path.exists(path):
        return None
    m = re.search(rb"CMDP_BUILD_ID=([0-9a-f]{64})", open(path, "rb").read())
    return m.group(1).decode() if m else None


def load():
    """Returns the loaded library; raises if it has not been built (python __graft_entry__.py / build())."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing: build the HIP extension first "
                f"(python -c 'import __graft_entry__ as g; g.build()').  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        L.cmdp_build_id.restype = C.c_char_p
        bid, want = L.cmdp_build_id().decode(), source_hash()
        if bid != want and os.environ.get("CMDP_ALLOW_STALE_LIB") != "1":
            raise ImportError(
                f"{LIB_PATH} was built from other sources (build id {bid[:16]}, tree {want[:16]}): rebuild it "
                f"(python -c 'import __graft_entry__ as g; g.build()').")
        vp, i32, i64, f32, f64 = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_double
        L.cmdp_version.restype = C.c_int
        L.cmdp_last_error.restype = C.c_char_p
        L.cmdp_device_count.restype = C.c_int
        L.cmdp_set_device.argtypes = [i32]
        L.cmdp_create.argtypes = [C.POINTER(vp), C.POINTER(CmdpDesc)]
        L.cmdp_destroy.argtypes = [vp]
        L.cmdp_stream.restype = vp
        L.cmdp_stream.argtypes = [vp]
        L.cmdp_reset.argtypes = [vp, vp, vp]
        L.cmdp_step.argtypes = [vp, vp, i32, vp, vp, vp]
        L.cmdp_rollout.argtypes = [vp, i32, vp, i64, vp, vp, vp, vp, vp]
        L.cmdp_rollout_async.argtypes = [vp, i32, i64]
        L.cmdp_synchronize.argtypes = [vp]
        L.cmdp_stat.argtypes = [vp, i32, vp]
        L.cmdp_calibrate.argtypes = [i32, i64, vp]
        L.cmdp_set_option.argtypes = [vp, i32, i64]
        L.cmdp_lds_plan.argtypes = [vp, vp]
        L.cmdp_visits.argtypes = [vp, vp, vp]
        L.cmdp_reset_visits.argtypes = [vp]
        L.cmdp_set_visits.argtypes = [vp, vp, vp]
        L.cmdp_state.argtypes = [vp, vp, vp, vp]
        L.cmdp_last_start.argtypes = [vp, vp, vp]
        L.cmdp_vi_discounted.argtypes = [vp, f32, f64, i32, i64, f64, vp, vp, vp, vp]
        L.cmdp_pe_discounted.argtypes = [vp, vp, f32, f64, i32, i64, vp, vp, vp, vp]
        L.cmdp_vi_episodic.argtypes = [vp, i32, vp, vp, vp]
        L.cmdp_pe_episodic.argtypes = [vp, i32, vp, vp, vp, vp]
        L.cmdp_diameter.argtypes = [vp, f64, i32, i64, vp, vp]
        L.cmdp_diameter_episodic.argtypes = [vp, i32, vp, vp, vp, f64, i64, vp, vp]
        L.cmdp_value_norm.argtypes = [vp, vp, vp]
        L.cmdp_gth.argtypes = [i32, vp, vp, vp]
        L.cmdp_qlearning_create.argtypes = [C.POINTER(vp), vp, vp, i64, f64, f64, f64, f64, i32]
        L.cmdp_qlearning_destroy.argtypes = [vp]
        L.cmdp_qlearning_run.argtypes = [vp, i64, vp, vp, vp]
        L.cmdp_qlearning_evaluate.argtypes = [vp, vp]
        L.cmdp_qlearning_run_logged.argtypes = [vp, C.POINTER(CmdpLoopDesc), i64, vp, vp, vp, vp, vp]
        L.cmdp_tracker_replay.argtypes = [C.POINTER(CmdpLoopDesc), i32, i32, vp, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
        L.cmdp_qlearning_continuous_create.argtypes = [C.POINTER(vp), vp, vp, i64, f64, f64, f64, f64]
        L.cmdp_qlearning_policy.argtypes = [vp, vp]
        L.cmdp_qlearning_average_reward.argtypes = [vp, vp, vp, vp]
        L.cmdp_average_reward.argtypes = [vp, vp, vp, vp, vp, vp, vp]
        L.cmdp_diameter_range.argtypes = [vp, C.c_double, i64, i64, i64, vp]
        L.cmdp_diameter_sparse_f64.argtypes = [vp, C.c_double, i64, vp, vp]
        L.cmdp_mixing_time.argtypes = [vp, vp, vp, C.c_double, i64, vp, vp]
        L.cmdp_set_observation_table.argtypes = [vp, vp, i32, i32]
        L.cmdp_observe.argtypes = [vp, C.c_double, vp]
        L.cmdp_observe_noise.argtypes = [vp, i32, C.c_double, C.c_double, vp, vp]
        L.cmdp_greedy_policy_episodic.argtypes = [vp, i32, i32, vp, vp]
        L.cmdp_qlearning_tables.argtypes = [vp, vp, vp]
        L.cmdp_set_reward_streams.argtypes = [vp, vp, vp, vp, vp]
        L.cmdp_legacy_beta.argtypes = [vp, vp, vp, vp, f64, f64, i64, vp]
        _lib = L
    return _lib


def check(rc):
    if rc == OK:
        return
    msg = load().cmdp_last_error().decode()
    if rc == ERR_MAX_ITER:
        raise DynamicProgrammingMaxIterationExceeded(msg)
    if rc == ERR_NEEDS_RESET:
        raise AssertionError(msg)  # the reference raises AssertionError (mdp/base.py:1290)
    raise CmdpError(rc, msg)


def ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def carr(a, dtype):
    return None if a is None else np.ascontiguousarray(a, dtype)


_hip = None


def pinned_empty(n, dtype):
    """A page-locked host array (hipHostMalloc) for results that are fetched repeatedly: device-to-host copies into it
    run as one DMA at PCIe rate and touch no fresh pages (a new `np.zeros` costs a page fault per 4 KiB inside the copy).
    Freed when the array is garbage collected."""
    import weakref

    global _hip
    load()  # libcmdp.so has mapped the HIP runtime
    if _hip is None:
        _hip = C.CDLL("libamdhip64.so")
        _hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]
        _hip.hipHostFree.argtypes = [C.c_void_p]
    dt = np.dtype(dtype)
    nbytes = max(1, int(n)) * dt.itemsize
    p = C.c_void_p()
    rc = _hip.hipHostMalloc(C.byref(p), nbytes, 0)
    if rc != 0 or not p.value:
        raise MemoryError(f"hipHostMalloc({nbytes}) failed with {rc}")
    buf = (C.c_char * nbytes).from_address(p.value)
    arr = np.frombuffer(buf, dtype=dt, count=int(n))
    weakref.finalize(buf, _hip.hipHostFree, C.c_void_p(p.value))
    return arr

"""ctypes binding of libmopoe_hip.so (the C ABI of include/mopoe_hip.h).

The library is the product: there is no PyTorch / CPU fallback.  If the shared
object is missing or exports the wrong ABI this module raises at import time.
"""
import ctypes as C
import os

# Host-side waits.  A training step here is ~30 us of GPU work: when the host does wait for the
# GPU (torch.cuda.synchronize, a blocking copy) ROCm's default -- sleep until an interrupt --
# costs 20-60 us per wait, polling ~5 us (HSA_ENABLE_INTERRUPT=0; the runtime reads it once, when
# it starts: set it -- or import this package / the mopoe_amd alias -- before the process's first
# HIP call; bench.py applies the same policy ahead of its own torch import).  The package asks for
# polling unless the user decided otherwise: an explicit
# HSA_ENABLE_INTERRUPT wins, MOPOE_HOST_WAIT=interrupt keeps ROCm's default.  bench.py and
# run_epochs.train therefore wait the same way (bench.py reports it as config.host_wait).
if os.environ.get("MOPOE_HOST_WAIT", "poll") != "interrupt":
    os.environ.setdefault("HSA_ENABLE_INTERRUPT", "0")

import torch  # noqa: E402,F401  (loads the ROCm runtime the library binds to)

MAX_MODS = 5
MAX_SUBSETS = 31
MAX_JOBS = 10
HIDDEN = 256
ROWS = 16
ABI_VERSION = 11
MAX_RANKS = 8
IPC_HANDLE_BYTES = 80
RCCL_ID_BYTES = 128
MAX_LAYERS = 4

SUB_POE, SUB_POE_PRIOR, SUB_SLICES = 0, 1, 2
JOINT_MIXTURE, JOINT_MEAN, JOINT_EXPERT = 0, 1, 2
LIKELIHOODS = {"normal": 0, "laplace": 1}      # MOPOE_LIK_*

STAT_TOTAL_LOSS = 0
STAT_JOINT_DIV = 1
STAT_KLD_SUBSET = 2
STAT_KLD_STYLE = 2 + MAX_SUBSETS
STAT_NLL = STAT_KLD_STYLE + MAX_MODS
STAT_LATENT_MEAN = STAT_NLL + MAX_JOBS    # + 4 * modality + {style mu, style lv, mu, lv}
NUM_STATS = STAT_LATENT_MEAN + 4 * MAX_MODS
STATS_ALLOC = 256          # floats of a stats buffer ([128:] diagnostic stamps)
NUM_COUNTERS = 64
COUNTERS_ALLOC = 128       # ints of a counters buffer ([64:] diagnostic stamps)
CTR_STEPS_BEGUN, CTR_STEPS_DONE, CTR_INVALID, CTR_FIRST_INVALID, CTR_ADAM_STEPS = 0, 1, 2, 3, 4
KERNEL_NAMES = ("k_linear", "k_latent", "k_wgrad", "k_adam", "k_finalize", "k_fused",
                "k_xgmi", "rccl_allreduce")

_i32 = C.c_int32
_u8 = C.c_uint8
_f32 = C.c_float
_ptr = C.c_void_p


class Model(C.Structure):
    _fields_ = [
        ("num_mods", _i32),
        ("class_dim", _i32),
        ("input_dim", _i32 * MAX_MODS),
        ("style_dim", _i32 * MAX_MODS),
        ("learn_output_scale", _i32),
        ("off_w1", _i32 * MAX_MODS),
        ("off_b1", _i32 * MAX_MODS),
        ("off_wh", _i32 * MAX_MODS),
        ("off_bh", _i32 * MAX_MODS),
        ("off_wd", _i32 * MAX_MODS),
        ("off_bd", _i32 * MAX_MODS),
        ("off_lvo", _i32 * MAX_MODS),
        ("off_ctrl", _i32),
        ("num_floats", _i32),
    ]


class Step(C.Structure):
    _fields_ = [
        ("n", _i32),
        ("present_mask", _i32),
        ("sample", _i32),
        ("joint_mode", _i32),
        ("expert_subset", _i32),
        ("backward", _i32),
        ("rows_per_group", _i32),
        ("group_rows", _i32),
        ("num_subsets", _i32),
        ("sub_mask", _i32 * MAX_SUBSETS),
        ("sub_avail", _i32 * MAX_SUBSETS),
        ("sub_kind", _i32 * MAX_SUBSETS),
        ("sub_members", (_i32 * MAX_MODS) * MAX_SUBSETS),
        ("sub_f", _i32 * MAX_SUBSETS),
        ("sub_kl_coef", _f32 * MAX_SUBSETS),
        ("num_comp", _i32),
        ("comp_sub", _i32 * MAX_SUBSETS),
        ("comp_f", _i32),
        ("comp_w", _f32 * MAX_SUBSETS),
        ("style_kl_coef", _f32 * MAX_MODS),
        ("num_jobs", _i32),
        ("job_mod", _i32 * MAX_JOBS),
        ("job_slot", _i32 * MAX_JOBS),
        ("job_src", _i32 * MAX_JOBS),
        ("job_stream", _i32 * MAX_JOBS),
        ("job_nll_coef", _f32 * MAX_JOBS),
        ("likelihood", _i32),
        ("job_eps_content", _ptr * MAX_JOBS),
        ("job_eps_style", _ptr * MAX_JOBS),
        ("seed", C.c_uint64),
        ("gemm_operands", _i32),
        ("pad_", _i32),
    ]


class Buffers(C.Structure):
    _fields_ = [
        ("params", _ptr),
        ("grads", _ptr),
        ("exp_avg", _ptr),
        ("exp_avg_sq", _ptr),
        ("counters", _ptr),
        ("x", _ptr * MAX_MODS),
        ("row_index", _ptr * MAX_MODS),
        ("x_rows", _i32 * MAX_MODS),
        ("hidden", _ptr * MAX_MODS),
        ("heads", _ptr * MAX_MODS),
        ("subsets_mu", _ptr),
        ("subsets_logvar", _ptr),
        ("joint_mu", _ptr),
        ("joint_logvar", _ptr),
        ("z", _ptr * MAX_MODS),
        ("loc", _ptr * MAX_MODS),
        ("stats", _ptr),
        ("stats_host", _ptr),
        ("status_host", _ptr),
        ("g_xhat", _ptr * MAX_MODS),
        ("g_heads", _ptr * MAX_MODS),
        ("g_pre", _ptr * MAX_MODS),
        ("wfrag", _ptr),
        ("partials", _ptr),
        ("wgrad_scratch", _ptr),
        ("wgrad_scratch_floats", C.c_int64),
    ]


class Topology(C.Structure):
    _fields_ = [
        ("enc_layers", _i32),
        ("dec_layers", _i32),
        ("dropout", _f32),
        ("sample_scale", _i32),
        ("off_we", (_i32 * MAX_LAYERS) * MAX_MODS),
        ("off_be", (_i32 * MAX_LAYERS) * MAX_MODS),
        ("off_wg", (_i32 * MAX_LAYERS) * MAX_MODS),
        ("off_bg", (_i32 * MAX_LAYERS) * MAX_MODS),
        ("off_wlv", _i32 * MAX_MODS),
        ("off_blv", _i32 * MAX_MODS),
    ]


class GBuffers(C.Structure):
    _fields_ = [
        ("enc_act", (_ptr * MAX_LAYERS) * MAX_MODS),
        ("enc_pre0", _ptr * MAX_MODS),
        ("dec_act", (_ptr * MAX_LAYERS) * MAX_MODS),
        ("g_enc", (_ptr * MAX_LAYERS) * MAX_MODS),
        ("g_dec", (_ptr * MAX_LAYERS) * MAX_MODS),
        ("lv", _ptr * MAX_MODS),
        ("g_lv", _ptr * MAX_MODS),
        ("g_z", _ptr * MAX_MODS),
        ("keep_enc", (_ptr * MAX_LAYERS) * MAX_MODS),
        ("keep_dec", (_ptr * MAX_LAYERS) * MAX_MODS),
    ]


class Adam(C.Structure):
    _fields_ = [("lr", _f32), ("beta1", _f32), ("beta2", _f32), ("eps", _f32)]


LIB_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)),
                        os.environ.get("MOPOE_LIB", "libmopoe_hip.so"))

# every symbol include/mopoe_hip.h declares: name -> (restype, argtypes)
SYMBOLS = {
    "mopoe_abi_version": (C.c_int, []),
    "mopoe_reload_knobs": (C.c_int, []),
    "mopoe_last_error": (C.c_char_p, []),
    "mopoe_sizeof": (C.c_int, [C.c_int]),
    "mopoe_profile_enable": (C.c_int, [C.c_int]),
    "mopoe_profile_read": (C.c_int, [C.POINTER(_i32), C.POINTER(_f32)]),
    "mopoe_model_layout": (C.c_int, [C.POINTER(Model)]),
    "mopoe_ldz": (C.c_int, [C.POINTER(Model), C.c_int]),
    "mopoe_partials_stride": (C.c_int, [C.POINTER(Model)]),
    "mopoe_wgrad_scratch_floats": (C.c_int64, [C.POINTER(Model), C.POINTER(Step)]),
    "mopoe_wfrag_floats": (C.c_int, [C.POINTER(Model)]),
    "mopoe_wfrag_refresh": (C.c_int, [C.POINTER(Model), C.POINTER(Buffers), C.c_void_p]),
    "mopoe_row_groups": (C.c_int, [C.POINTER(Model), C.POINTER(Step)]),
    "mopoe_latent_lds_bytes": (C.c_int, [C.POINTER(Model), C.POINTER(Step)]),
    "mopoe_forward": (C.c_int, [C.POINTER(Model), C.POINTER(Step),
                                C.POINTER(Buffers), _ptr]),
    "mopoe_train_step": (C.c_int, [C.POINTER(Model), C.POINTER(Step),
                                   C.POINTER(Buffers), C.POINTER(Adam), _ptr]),
    "mopoe_adam_step": (C.c_int, [C.POINTER(Model), _i32, C.POINTER(Buffers),
                                  C.POINTER(Adam), _i32, _ptr]),
    "mopoe_comm_create": (C.c_int, [_i32, _i32, _i32, _i32, C.POINTER(_ptr), _ptr]),
    "mopoe_comm_connect": (C.c_int, [_ptr, _ptr]),
    "mopoe_comm_allreduce": (C.c_int, [_ptr, _ptr, _ptr]),
    "mopoe_comm_allreduce_adam": (C.c_int, [_ptr, C.POINTER(Model), _i32,
                                            C.POINTER(Buffers), C.POINTER(Adam), _ptr]),
    "mopoe_comm_train_step": (C.c_int, [_ptr, C.POINTER(Model), C.POINTER(Step),
                                        C.POINTER(Buffers), C.POINTER(Adam), _ptr]),
    "mopoe_comm_status": (C.c_int, [_ptr, C.POINTER(_i32)]),
    "mopoe_comm_destroy": (C.c_int, [_ptr]),
    "mopoe_topology_layout": (C.c_int, [C.POINTER(Model), C.POINTER(Topology)]),
    "mopoe_general_enc_blocks": (C.c_int, [C.POINTER(Topology), C.POINTER(Step), C.c_int]),
    "mopoe_general_forward": (C.c_int, [C.POINTER(Model), C.POINTER(Topology), C.POINTER(Step),
                                        C.POINTER(Buffers), C.POINTER(GBuffers), _ptr]),
    "mopoe_general_train_step": (C.c_int, [C.POINTER(Model), C.POINTER(Topology),
                                           C.POINTER(Step), C.POINTER(Buffers),
                                           C.POINTER(GBuffers), C.POINTER(Adam), _ptr, _ptr]),
    "mopoe_general_adam_step": (C.c_int, [C.POINTER(Model), C.POINTER(Topology), _i32,
                                          C.POINTER(Buffers), C.POINTER(Adam), _i32, _ptr]),
    "mopoe_sampler_epoch": (C.c_int, [_ptr, C.POINTER(_i32), _i32, _ptr, _ptr, C.c_int64, _ptr,
                                      _ptr, _ptr, C.POINTER(C.c_int64)]),
    "mopoe_sampler_rows": (C.c_int, [_i32, C.c_int64, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr, _ptr,
                                     _ptr]),
    "mopoe_rccl_unique_id": (C.c_int, [_ptr]),
    "mopoe_rccl_create": (C.c_int, [_i32, _i32, _ptr, C.POINTER(_ptr)]),
    "mopoe_rccl_train_step": (C.c_int, [_ptr, C.POINTER(Model), C.POINTER(Step),
                                        C.POINTER(Buffers), C.POINTER(Adam), _ptr]),
    "mopoe_rccl_allreduce": (C.c_int, [_ptr, _ptr, C.c_int64, _ptr]),
    "mopoe_rccl_info": (C.c_int, [_ptr, C.POINTER(_i32), C.POINTER(_i32)]),
    "mopoe_rccl_destroy": (C.c_int, [_ptr]),
    "mopoe_linear": (C.c_int, [_ptr, _i32, _i32, _ptr, _ptr, _i32, _i32, _ptr,
                               _ptr]),
    "mopoe_poe": (C.c_int, [_ptr, _ptr, _i32, C.c_int64, _f32, _ptr, _ptr,
                            _ptr]),
    "mopoe_kl_divergence": (C.c_int, [_ptr, _ptr, C.c_int64, _f32, _ptr, _ptr,
                                      _ptr]),
    "mopoe_reparameterize": (C.c_int, [_ptr, _ptr, _ptr, C.c_int64,
                                       C.c_uint64, C.c_uint64, _ptr, _ptr]),
    "mopoe_mixture_select": (C.c_int, [_ptr, _ptr, _i32, _i32, _i32, _ptr,
                                       _ptr, _ptr, _ptr]),
}


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            "libmopoe_hip.so is missing (%s): build it with "
            "`make -C 2022_cambroise_interpret_multivae_amd/csrc` or "
            "`python -c 'import __graft_entry__ as g; g.build()'`. There is "
            "no CPU fallback." % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (restype, argtypes) in SYMBOLS.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is absent
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.mopoe_abi_version() != ABI_VERSION:
        raise ImportError("libmopoe_hip.so ABI %d != binding ABI %d"
                          % (lib.mopoe_abi_version(), ABI_VERSION))
    mirrors = [C.sizeof(Model), C.sizeof(Step), C.sizeof(Buffers),
               C.sizeof(Adam), Step.job_eps_content.offset,
               Step.comp_w.offset, Buffers.partials.offset,
               Model.num_floats.offset, Buffers.status_host.offset,
               Model.off_ctrl.offset, Buffers.wfrag.offset, C.sizeof(Topology),
               C.sizeof(GBuffers), GBuffers.keep_enc.offset]
    for which, mine in enumerate(mirrors):
        if lib.mopoe_sizeof(which) != mine:
            raise ImportError("ctypes mirror %d disagrees with the C struct "
                              "(%d vs %d)" % (which, mine,
                                              lib.mopoe_sizeof(which)))
    return lib


lib = _load()


class MopoeError(RuntimeError):
    pass


def check(rc, what):
    if rc != 0:
        raise MopoeError("%s failed (%d): %s" % (
            what, rc, lib.mopoe_last_error().decode("utf-8", "replace")))


def rows_per_group():
    """Step.rows_per_group the plans are built with: 0 (the library picks) unless
    MOPOE_ROWS_PER_GROUP pins it (a test / tuning knob; results do not depend on it
    beyond the summation order of the scalar partials)."""
    return int(os.environ.get("MOPOE_ROWS_PER_GROUP", "0"))


def require_gpu(t=None):
    """The product path runs on the MI355X only."""
    if not torch.cuda.is_available():
        raise MopoeError("no HIP device visible: the MoPoE hot path has no "
                         "CPU fallback")
    if t is not None and not t.is_cuda:
        raise MopoeError("expected a device tensor, got %s" % t.device)


def device_rows(t, device=None):
    """`t` as a contiguous float32 matrix on `device` (itself when it already is one).
    The kernels read an input matrix through a descriptor of exactly rows * d floats:
    nothing beyond the tensor is touched, whatever its row length or where its storage
    ends (the 16-byte load over the tail of the last row is masked by the hardware's
    per-dword range check)."""
    device = torch.device(device) if device is not None else t.device
    if device.type == "cuda" and device.index is None:
        device = torch.device("cuda", torch.cuda.current_device())
    if t.device == device and t.dtype == torch.float32 and t.is_contiguous():
        return t
    return t.to(device=device, dtype=torch.float32).contiguous()


def reload_knobs():
    """Re-read the library's environment knobs (it reads them once, when it is loaded)."""
    check(lib.mopoe_reload_knobs(), "mopoe_reload_knobs")


def profile_enable(on):
    check(lib.mopoe_profile_enable(int(bool(on))), "mopoe_profile_enable")


def profile_read():
    """{kernel name: (launches, total ms)} since the last read."""
    n = len(KERNEL_NAMES)
    count = (_i32 * n)()
    ms = (_f32 * n)()
    check(lib.mopoe_profile_read(count, ms), "mopoe_profile_read")
    return {KERNEL_NAMES[k]: (int(count[k]), float(ms[k])) for k in range(n)}


def stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())

"""Host-side launch plan for the OTVM frame path on MI355X.

The reference executes ``EvalModel.forward`` (models/alpha/model.py:391-512) as ~600 ATen calls per
frame.  Here the same function is compiled once per input resolution into a static list of HIP kernel
launches over pre-allocated NHWC buffers (``FramePlan``); running a frame is a tight loop of ctypes calls
into ``libotvm_hip.so`` on torch's current HIP stream.  PyTorch is used for device memory only.

What is decided at plan time (load time for weights):
  * weight standardisation of the 66 WS convs is applied once (``otvm_pack_conv_weight(ws=1)``),
    instead of per forward as ``layers_WS.py:15-21`` does;
  * the 86 eval-mode BatchNorms of the STM encoders are folded into their convolutions;
  * the five stem convolutions of ``Encoder_M`` (STM.py:56-66) become one 24-channel convolution;
  * every ``torch.cat`` is replaced by producers writing channel slices of the consumer's buffer;
  * the memory bank is a set of per-slot key/value buffers (``alpha/model.py:472-493`` policy kept).
"""
import ctypes as C
import os

import torch

from . import lib as L

NONE, RELU, LEAKY = 0, 1, 2
GUARD_CLEAR = 2 ** 31 - 1      # value of the range-guard flag while nothing was flagged (otvm_finite_guard keeps the minimum tag)
F16_LIMIT = 65504.0
FUSE_GN_STATS = True       # GroupNorm statistics accumulated in the producing conv's epilogue
# GroupNorm apply folded into the staging of the ONLY consumer when that is a patch conv (the normalised tensor is never
# written: refinement BasicBlocks at full resolution, FBA layer1); OTVM_FUSE_GN_APPLY=0 keeps the separate pass
FUSE_GN_APPLY = os.environ.get("OTVM_FUSE_GN_APPLY", "1") != "0"
# OTVM_GRAPHS=1: static launch lists replayed as hipGraphs.  Off by default: every configuration measured is GPU-bound
# (1080p 37.6 vs 37.6 fps, 480p 124.9 vs 123.9, IO pipeline 37.0 vs 36.9), the graphs only cut the host time per frame
# (6.8 -> 0.9 ms at 480p) -- worth switching on when many processes share few host cores.
# Round 3: "auto" (OTVM_GRAPHS unset) replays graphs when the host is the scarce resource -- several ranks share this node's
# cores (WORLD_SIZE > 1: eight ranks issuing ~320 launches per frame each from Python), or the frame is so small that the
# host's issue time (6-7 ms) is of the order of the device time (padded frame below 2^20 pixels: 480p runs 7 ms frames).
_GRAPHS_ENV = os.environ.get("OTVM_GRAPHS")
USE_GRAPHS = None if _GRAPHS_ENV in (None, "", "auto") else (_GRAPHS_ENV != "0")
# (end of round 3: 1080p too -- 43.78 vs 43.60 frames/s, host issue time 20.6 -> 3.4 ms per frame; 4K frames take > 100 ms, direct launches)
GRAPH_AUTO_PIXELS = 1 << 22


def graphs_wanted(setting, padded_pixels):
    """Resolve engine.use_graphs (True / False / None = auto) for a plan of ``padded_pixels``."""
    if setting is not None:
        return bool(setting)
    if int(os.environ.get("WORLD_SIZE", "1")) > 1:
        return True
    return padded_pixels < GRAPH_AUTO_PIXELS
PRE_ON_S2 = os.environ.get("OTVM_PRE_ON_S2", "1") != "0"           # preprocess + statistics clear on the second side stream
EVDEC_LATE = os.environ.get("OTVM_EVDEC_LATE", "1") != "0"         # release the next frame's query encoder behind the trimap encoding
# one-wave 64x64 tile (operands straight from L2 into MFMA registers, no LDS / barriers in the K loop) as an autotuner
# candidate.  Built and verified in round 3, measured 20-90 % SLOWER than the 4-wave LDS tiles on every small-map shape
# (scattered 32-byte A loads: 32 cache lines per load instruction through a 64 B/clk L1): off by default, no weight copy.
WAVE_TILE = os.environ.get("OTVM_WAVE_TILE", "0") != "0"
# round 3: GroupNorm apply folded into the staging of implicit-GEMM convs too (bn2 -> conv3 of every FBA bottleneck); 0 = only
# into the 3x3 patch kernel, as in round 2
FUSE_GN_APPLY_IGEMM = os.environ.get("OTVM_FUSE_GN_APPLY_IGEMM", "1") != "0"
FUSE_GN_APPLY_IGEMM_KXK = os.environ.get("OTVM_FUSE_GN_APPLY_IGEMM_KXK", "0") != "0"   # also into 3x3 implicit-GEMM layers (measured: a loss)
# round 3 (ABI 16): the GroupNorm scale / shift table of a conv's output is written by that conv's last workgroup; 0 = one
# otvm_gn_table launch per table (42 per frame)
FUSE_GN_TABLE = os.environ.get("OTVM_FUSE_GN_TABLE", "1") != "0"
# round 3: each 1/4-resolution bottleneck of the STM encoders (res2.0-2, planes 64) as ONE kernel, intermediates in LDS
# (csrc/bottleneck_f16x3.hip); f16x3 only.  0 = the three (four) convolution launches of round 2
FUSE_STM_BLOCK = os.environ.get("OTVM_FUSE_STM_BLOCK", "1") != "0"
# round 6 (ABI 19): the identity bottlenecks of the STM encoders' 1/8-resolution stage (res3.1 - res3.3, planes 128) as ONE kernel
# too (csrc/bottleneck128_f16x3.hip).  Whether a block runs fused -- and on which pixel tile -- or as its three convolution
# launches is TIMED at plan time per map size (FramePlan._resolve_stm128; cached with the conv tuner's choices).
# OTVM_FUSE_STM_BLOCK128: 0 = never, 1 = timed (default), 2 = always (tile from the map size)
FUSE_STM_BLOCK128 = int(os.environ.get("OTVM_FUSE_STM_BLOCK128", "1"))
STM128_TILE = int(os.environ.get("OTVM_STM128_TILE", "0"))       # (experiments, with OTVM_FUSE_STM_BLOCK128=2: force pixel tile 1 / 2 / 3)
FUSE_PPM_HEAD = os.environ.get("OTVM_PPM_HEAD", "1") != "0"
# round 4 (ABI 17): the GroupNorm statistics of conv3's OUTPUT in the FBA bottlenecks predicted from its input (channel sums +
# Gram matrix, csrc/gram.hip), so conv3's epilogue normalises, adds the identity, applies the ReLU and writes the block output:
# the bn3 apply pass (read raw output + identity, write the block output: 16 passes, 1.25 ms per 1080p frame) is gone.
# OTVM_GN_PREDICT=0 = round 3's apply passes; OTVM_GN_PREDICT_PASSES=3 = Gram matrix on f16x3 operands instead of fp16;
# OTVM_GN_PREDICT_DS=0 keeps the apply pass of the four blocks with a projection (their identity is a raw GroupNorm input:
# conv3's epilogue scales it per channel, otvm_conv_params.res_scale)
FUSE_GN_PREDICT = os.environ.get("OTVM_GN_PREDICT", "1") != "0"
# ... on maps of at least this many pixels (the 1/8-resolution maps of a 1080p frame hold 32 640): below, the two extra launches
# per block cost more than the pass they replace, and the fp16 pass's noise on the statistics (~ 1 / sqrt(pixels)) grows
GN_PREDICT_MIN_PIXELS = int(os.environ.get("OTVM_GN_PREDICT_MIN_PIXELS", "16384"))
GN_PREDICT_PASSES = int(os.environ.get("OTVM_GN_PREDICT_PASSES", "1"))
# ... and for these bottleneck widths (planes of conv3's INPUT; "64,128,256,512" = all four stages of the FBA encoder)
GN_PREDICT_PLANES = tuple(int(v) for v in os.environ.get("OTVM_GN_PREDICT_PLANES", "64,128,256,512").split(",") if v)
GN_PREDICT_DS = os.environ.get("OTVM_GN_PREDICT_DS", "1") != "0"
# round 5 (ABI 18): conditioning guard of the predicted statistics.  var = E[y^2] - mean^2 receives the Gram matrix's rounding error
# (fp16 operands: ~3e-7 of E[y^2] after averaging) amplified by 1 + kappa, kappa = mean^2 / var of the output group.  The
# prediction kernel keeps the running maximum of kappa per layer (and flags non-finite statistics / saturated fp16 operands); the
# host reads it after the first frame of every clip (the caller synchronises there anyway) and every few frames in between:
#   kappa > KAPPA_P3  -> that layer's Gram matrix is computed on f16x3 operands from now on (22-bit operands: error ~1e-9 of E[y^2])
#   kappa > KAPPA_OFF, non-finite, saturated -> the prediction is switched off for this engine: the plans are rebuilt on the
#                        accumulated route (statistics summed from conv3's own output, round 3) at the next clip boundary
# and a first frame whose statistics tripped either threshold is computed again before it is returned.
# Measured (tests/test_gpu_kernels.py::test_gn_predict_conditioning_guard, kappa = 1667): relative error of rstd 4.6e-4 with the
# fp16 Gram matrix, 8.2e-6 with the f16x3 one, i.e. (1 + kappa) x 2.8e-7 and (1 + kappa) x 4.9e-9 -- the thresholds keep it
# below ~1.5e-6 in every regime (the accumulated route itself is at 1e-7 .. 1e-6).
GN_PREDICT_KAPPA_P3 = float(os.environ.get("OTVM_GN_PREDICT_KAPPA_P3", "4"))
GN_PREDICT_KAPPA_OFF = float(os.environ.get("OTVM_GN_PREDICT_KAPPA_OFF", "256"))
# round 4 (ABI 17): the refinement's last BasicBlock ends in bn2 -> (+ identity) -> ReLU with ONE reader, pred.0 (a 3x3 patch
# conv): its staging normalises, adds the identity and applies the ReLU (otvm_conv_params.in_res) -- the 535 MB block output
# is never written (one 1.6 GB apply pass less per frame, 535 MB more read by pred.0).  0 = round 3's apply pass
FUSE_REFINE_TAIL = os.environ.get("OTVM_FUSE_REFINE_TAIL", "0") != "0"
# round 4 (ABI 17): conv_up4.2 / pred.2 (3x3, 32 -> 16) carry the 1x1 head + fba_fusion that follows them in their epilogue
# (otvm_conv2d_head): the decoder's hidden state is not written at all (unless a training forward needs it), the
# refinement's is written once and not read back; two launches per frame less.  0 = conv + otvm_fba_head
FUSE_HEAD = os.environ.get("OTVM_FUSE_HEAD", "1") != "0"
HEAD16 = os.environ.get("OTVM_HEAD16", "1") != "0"
# (round 4, measured and rejected, code removed in round 5: the PPM chain -- pooling, heads, Z table: ~110 us of small launches that need
# layer 4 only -- on an auxiliary stream BESIDE conv_up1.0's layer-4 part instead of in front of it: 43.8 / 43.7 frames/s against
# 44.5 / 44.5 serial on one box; the pooling is an HBM-speed pass and takes from the convolution what it gives.  HISTORY.md)
# round 3: the PPM branches' third of conv_up1.0 computed from the 50 pooled pixels (otvm_ppm_conv_z / _add) instead of
# convolving their upsampled copies; conv_up1.0 then reads layer 4 only.  OTVM_PPM_ALGEBRA=0 keeps the materialised form.
PPM_ALGEBRA = os.environ.get("OTVM_PPM_ALGEBRA", "1") != "0"       # the four PPM heads in one launch (otvm_ppm_head)


# Plan-time autotuning of the convolution configurations (f16x3): every distinct layer shape is timed once on the device
# over the legal (kernel, tile, K-split) configurations otvm_conv2d_candidates lists, the fastest is kept in
# otvm_conv_params.tune.  All configurations compute the same convolution (fp32 summation order differs).  The choice is
# cached per process and per layer signature, so two plans / engines of one process run identical configurations.
# OTVM_AUTOTUNE=0 keeps the built-in heuristic (thresholds tuned by hand on the 1080p frame).
AUTOTUNE = os.environ.get("OTVM_AUTOTUNE", "1") != "0"
_TUNE_CACHE = {}
TUNE_LOG = []           # (signature, chosen code, {code: ms}) of every shape timed in this process (tools / DESIGN numbers)
# OTVM_TUNE_FILE=path: choices are loaded from / saved to a JSON file, so a later process (a profiler run, a service
# restart) launches the tuned configurations without timing anything.
TUNE_FILE = os.environ.get("OTVM_TUNE_FILE")
_TUNE_FILE_LOADED = False


# OTVM_* variables that do NOT select kernels, tiles or launch routes (everything else with that prefix does)
_NON_VARIANT_ENV = ("OTVM_TUNE_FILE", "OTVM_DIST_BACKEND", "OTVM_TEST_FULL_F64", "OTVM_TEST_4K_ORACLE", "OTVM_BENCH_LIVE_PMC",
                    "OTVM_BENCH_FORCE_DIAG", "OTVM_CHECK_FINITE", "OTVM_GRAPHS")


def variant_env():
    """The OTVM_* switches of this process that select kernel variants (engine routes; with the -DOTVM_PROBES library also tile
    thresholds / kernel forms / tile walks), sorted."""
    return sorted((k, v) for k, v in os.environ.items() if k.startswith("OTVM_") and k not in _NON_VARIANT_ENV)


def kernel_config_digest():
    """16 hex digits over everything that selects kernel variants in this process: the tuned configuration of every layer shape
    AND the OTVM_* environment switches the library and the engine read (tile thresholds, kernel-variant switches such as
    OTVM_PATCH_WIDE_NWN or OTVM_IGEMM_GLDS change fp32 summation orders without passing through the tuner).  Equal
    digests on all ranks = identical launches on all ranks.  (A conditioning-guard trip -- HipEngine.gn_predict_log -- rebuilds
    that rank's plans on another route and may time new layer signatures: digests of ranks can differ afterwards.)"""
    import hashlib
    return hashlib.sha256(repr((sorted(_TUNE_CACHE.items()), variant_env())).encode()).hexdigest()[:16]


def _tune_file_tag():
    """What a tune file is valid for: the chip, the library ABI and the kernel-variant switches of the process that timed it (a
    tune code of 0 resolves to different kernels under different switches: choices timed under one setting must not be applied
    silently under another -- ADVICE r5)."""
    # (not the marketing name: it comes from an ids file that is missing on some boxes and was seen to differ between two
    # processes of one box -- the ISA name and the CU count identify the chip)
    name = "none"
    if torch.cuda.is_available():
        pr = torch.cuda.get_device_properties(torch.cuda.current_device())
        name = "%s/%dcu" % (getattr(pr, "gcnArchName", "?").split(":")[0], pr.multi_processor_count)
    return {"device": name, "abi": L.ABI_VERSION, "variant_env": [list(kv) for kv in variant_env()]}


def _load_tune_file():
    """Never raises: a missing, truncated or foreign file is ignored with a warning (the shapes are timed again)."""
    if not (TUNE_FILE and os.path.exists(TUNE_FILE)):
        return
    import json
    import warnings
    try:
        with open(TUNE_FILE) as f:
            doc = json.load(f)
        if not isinstance(doc, dict) or "choices" not in doc:
            raise ValueError("no 'choices' table (file written by an older build)")
        tag = _tune_file_tag()
        if doc.get("abi") != tag["abi"] or (tag["device"] != "none" and doc.get("device") != tag["device"]):
            raise ValueError("tuned for %r / ABI %r, this process runs %r / ABI %r"
                             % (doc.get("device"), doc.get("abi"), tag["device"], tag["abi"]))
        if doc.get("variant_env", []) != tag["variant_env"]:
            raise ValueError("tuned under the switches %r, this process runs under %r" % (doc.get("variant_env", []), tag["variant_env"]))
        for k, v in doc["choices"].items():
            _TUNE_CACHE[tuple(json.loads(k))] = int(v)
    except Exception as e:                                       # noqa: BLE001 -- any defect of the file means "not usable"
        warnings.warn("otvm_amd: ignoring OTVM_TUNE_FILE=%s (%s)" % (TUNE_FILE, e))


def _save_tune_file():
    """Atomic (temp file + rename): ranks of one --gpus N run share the environment and may save concurrently."""
    if not TUNE_FILE:
        return
    import json
    doc = dict(_tune_file_tag())
    doc["choices"] = {json.dumps([int(x) for x in k]): v for k, v in _TUNE_CACHE.items()}
    tmp = "%s.%d.tmp" % (TUNE_FILE, os.getpid())
    with open(tmp, "w") as f:
        json.dump(doc, f, indent=0)
    os.replace(tmp, TUNE_FILE)


def share_tune_cache(src=0):
    """Multi-rank runs: every rank adopts rank ``src``'s choices (broadcast over the initialised process group), so all
    ranks launch identical kernel configurations -- identical fp32 summation orders -- and a clip's alpha does not depend
    on the rank that matted it.  Call it after ``src`` has built the plans of the resolutions in play (bench.py, eval_cli)
    and before the other ranks build theirs.  No-op without a process group."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return                                               # (a one-rank group still broadcasts: the RCCL path is exercised)
    obj = [dict(_TUNE_CACHE) if dist.get_rank() == src else None]
    dist.broadcast_object_list(obj, src=src)
    if dist.get_rank() != src:
        _TUNE_CACHE.update(obj[0])


def _rup(x, m):
    return (x + m - 1) // m * m


class Act:
    """A [H, W, C] fp32 NHWC view with pixel stride ``ld`` inside a flat device buffer -- or a batch of ``B`` such views,
    image b lying ``bs`` elements behind image 0 (independent sequences stepped in lock-step, one launch per layer)."""
    __slots__ = ("t", "H", "W", "C", "ld", "off", "B", "bs")

    def __init__(self, t, H, W, C, ld=None, off=0, B=1, bs=0):
        self.t, self.H, self.W, self.C = t, H, W, C
        self.ld = C if ld is None else ld
        self.off = off
        self.B, self.bs = B, (bs if B > 1 else 0)

    @property
    def ptr(self):
        return self.t.data_ptr() + 4 * self.off

    @property
    def P(self):
        return self.H * self.W

    def ch(self, c0, c):
        return Act(self.t, self.H, self.W, c, self.ld, self.off + c0, self.B, self.bs)

    def img(self, b):
        """Image b of the batch as a single-image view."""
        return Act(self.t, self.H, self.W, self.C, self.ld, self.off + b * self.bs)

    def torch(self, b=0):
        """[H, W, C] strided torch view of image b (tests / debugging only)."""
        return torch.as_strided(self.t, (self.H, self.W, self.C), (self.W * self.ld, self.ld, 1), self.off + b * self.bs)


class ConvW:
    __slots__ = ("w", "K_pad", "O", "I", "I_pad", "kh", "kw", "bias", "w_hi", "w_lo", "w_scale", "w_frag", "w_wfrag", "w16")


def pad_amounts(h, w, d):
    """(lw, uw, lh, uh): reference models/alpha/common.py:6-27."""
    nh = h + (d - h % d) % d
    nw = w + (d - w % d) % d
    lh, lw = int((nh - h) / 2), int((nw - w) / 2)
    return lw, nw - w - lw, lh, nh - h - lh


def bank_update(bank, new, first_frame, memorize, max_memory_num):
    """Slot policy of reference models/alpha/model.py:472-493.  Returns (bank, released_slots)."""
    old = list(bank)
    if max_memory_num == 0:
        nb = [new] if first_frame else bank
    elif max_memory_num == 1:
        nb = [new]
    else:
        if first_frame:
            nb = [new]
        elif memorize or len(bank) == 1:
            nb = bank + [new]
        else:
            nb = bank[:-1] + [new]
        if len(nb) > max_memory_num:
            nb = nb[:1] + nb[2:]
    released = [s for s in old + [new] if not any(s is k for k in nb)]
    return nb, released


# name -> (kernel family, otvm_conv_params.precision).  "f16" (round 5) is a LABELLED reduced-precision mode: the f16x3 kernels,
# weight formats and plans with ONE MFMA pass on fp16-rounded operands in the implicit-GEMM and patch kernels (include/otvm_hip.h,
# OTVM_PREC_F16); never a default, not covered by the 1e-3 contract -- bench.py --precision f16 reports its error next to its speed
PRECISIONS = {"f32": L.PREC_F32, "f16x3": L.PREC_F16X3, "f16": L.PREC_F16X3}
CONV_PRECISIONS = {"f32": L.PREC_F32, "f16x3": L.PREC_F16X3, "f16": L.PREC_F16}


def default_precision():
    """f16x3 (split-fp16 MFMA, fp32-class accuracy) unless OTVM_PRECISION=f32 asks for the exact-fp32 MFMA."""
    import os
    return os.environ.get("OTVM_PRECISION", "f16x3")


def pack_conv_weight(lib, dev, w, ws=False, scale=None, i_pad=None, split=True, stream=0):
    """OIHW fp32 weight -> ConvW (packed fp32 + optional f16x3 split) on ``dev``."""
    O, I, kh, kw = w.shape
    cw = ConvW()
    cw.O, cw.I, cw.kh, cw.kw = O, I, kh, kw
    cw.I_pad = _rup(I, 4) if i_pad is None else i_pad
    cw.K_pad = _rup(kh * kw * cw.I_pad, 32)
    O_pad = _rup(O, 128)
    cw.w = torch.empty(O_pad * cw.K_pad, dtype=torch.float32, device=dev)
    cw.bias = None
    w = w.contiguous()
    L.check(lib.otvm_pack_conv_weight(w.data_ptr(), O, I, kh, kw, 1 if ws else 0,
                                      0 if scale is None else scale.data_ptr(), cw.w.data_ptr(), O_pad,
                                      cw.I_pad, cw.K_pad, stream), "pack_conv_weight")
    cw.w_hi = cw.w_lo = cw.w_scale = cw.w_frag = cw.w_wfrag = cw.w16 = None
    if split:
        cw.w_hi = torch.empty(O_pad * cw.K_pad, dtype=torch.float16, device=dev)
        cw.w_lo = torch.empty(O_pad * cw.K_pad, dtype=torch.float16, device=dev)
        cw.w_scale = torch.empty(O, dtype=torch.float32, device=dev)
        L.check(lib.otvm_split_conv_weight_f16x3(cw.w.data_ptr(), O, O_pad, cw.K_pad, kh * kw, cw.I_pad, cw.w_hi.data_ptr(),
                                                 cw.w_lo.data_ptr(), cw.w_scale.data_ptr(), stream), "split_conv_weight")
        # whole 32-channel chunks: the fragment-major copy -- what the LDS-DMA weight stages of the 256-row implicit-GEMM tiles
        # copy (round 5: layers with at least 128 filters), and what the one-wave tile reads (OTVM_WAVE_TILE)
        if cw.I_pad % 32 == 0 and kh * kw <= 32 and (WAVE_TILE or O >= 128):
            cw.w_wfrag = torch.zeros(int(lib.otvm_wave_weight_bytes_f16x3(O_pad, cw.K_pad)), dtype=torch.uint8, device=dev)
            L.check(lib.otvm_pack_wave_weight_f16x3(cw.w_hi.data_ptr(), cw.w_lo.data_ptr(), O_pad, cw.K_pad, cw.w_wfrag.data_ptr(),
                                                    stream), "pack_wave_weight")
        if kh == 7 and kw == 7 and O <= 64 and cw.I_pad <= 64:   # 7x7 stems: fragment-major copy for the stem kernel
            cw.w_frag = torch.zeros(int(lib.otvm_stem_weight_bytes_f16x3(cw.I_pad)), dtype=torch.uint8, device=dev)
            L.check(lib.otvm_pack_stem_weight_f16x3(cw.w.data_ptr(), O, cw.K_pad, cw.I_pad, cw.w_frag.data_ptr(),
                                                    cw.w_scale.data_ptr(), stream), "pack_stem_weight")
        if kh == 3 and kw == 3 and cw.I_pad % 16 == 0:          # 3x3: also the fragment-major copy for the patch kernel
            cw.w_frag = torch.zeros(int(lib.otvm_patch_weight_bytes_f16x3(O, cw.I_pad)), dtype=torch.uint8, device=dev)
            L.check(lib.otvm_pack_patch_weight_f16x3(cw.w.data_ptr(), O, cw.K_pad, cw.I_pad, cw.w_frag.data_ptr(),
                                                     cw.w_scale.data_ptr(), stream), "pack_patch_weight")
            if O == 16 and cw.I_pad == 32:                        # ... and as B fragments of the 16-wide tile (otvm_conv2d_head)
                cw.w16 = torch.zeros(int(lib.otvm_head16_weight_bytes_f16x3()), dtype=torch.uint8, device=dev)
                L.check(lib.otvm_pack_head16_weight_f16x3(cw.w.data_ptr(), O, cw.K_pad, cw.I_pad, cw.w_scale.data_ptr(),
                                                          cw.w16.data_ptr(), stream), "pack_head16_weight")
    return cw


def gram_tables(lib, cw):
    """(Mp fp32 [32][entries], v fp64 [32][Cin]) of a bias-free 1x1 conv for otvm_gn_predict (csrc/gram.hip): per GroupNorm
    group g of its OUTPUT channels v_g = sum of the group's filters and M_g = sum of their outer products, from the packed
    (standardised) fp32 weights in float64; M in the block-upper-triangular order of the Gram kernel's partials, off-diagonal
    blocks doubled (the symmetric half they stand for)."""
    O_pad = cw.w.numel() // cw.K_pad
    w = cw.w.view(O_pad, cw.K_pad)[:cw.O, :cw.I].double()               # 1x1: K index = input channel
    cg, Cin = cw.O // 32, cw.I
    wg = w.view(32, cg, Cin)
    v = wg.sum(1).contiguous()                                           # [32][Cin]
    bs = int(lib.otvm_gram_block(Cin))
    nb = Cin // bs
    blocks = []
    for bi in range(nb):
        rows = wg[:, :, bi * bs:(bi + 1) * bs]
        for bj in range(bi, nb):
            m = torch.einsum("gci,gcj->gij", rows, wg[:, :, bj * bs:(bj + 1) * bs])
            blocks.append((m if bi == bj else 2.0 * m).reshape(32, bs * bs))
    mp = torch.cat(blocks, 1).float().contiguous()                       # [32][nblk * bs * bs]: group-major
    assert mp.shape[1] == int(lib.otvm_gram_entries(Cin))
    return mp, v


def conv_params(x, cw, out, bias=None, stride=1, pad=0, dil=1, act=NONE, in_relu=0, residual=None, precision=L.PREC_F32,
                in_norm=None, splitk_ws=None):
    """in_norm = (scale_ptr, shift_ptr, act[, floats between the images' tables]): fused normalisation of the input
    (otvm_conv_params.in_scale); a batched input (x.B > 1) makes the launch a batched one;
    splitk_ws = float tensor the library may use for split-K partial tiles (one per concurrently used stream)."""
    Ho = (x.H + 2 * pad - dil * (cw.kh - 1) - 1) // stride + 1
    Wo = (x.W + 2 * pad - dil * (cw.kw - 1) - 1) // stride + 1
    if precision in (L.PREC_F16X3, L.PREC_F16) and cw.w_hi is None:
        raise RuntimeError("otvm_amd: f16x3 convolution requested but the weight was packed without a split")
    return L.ConvParams(x.ptr, x.H, x.W, x.C, x.ld, cw.w.data_ptr(), cw.K_pad,
                        0 if bias is None else bias.data_ptr(),
                        0 if residual is None else residual.ptr, 0 if residual is None else residual.ld,
                        out.ptr, Ho, Wo, cw.O, out.ld, cw.kh, cw.kw, stride, pad, dil, in_relu, act, precision,
                        0 if cw.w_hi is None else cw.w_hi.data_ptr(), 0 if cw.w_lo is None else cw.w_lo.data_ptr(),
                        0 if cw.w_scale is None else cw.w_scale.data_ptr(),
                        0 if cw.w_frag is None else cw.w_frag.data_ptr(), 0,
                        0 if in_norm is None else in_norm[0], 0 if in_norm is None else in_norm[1],
                        0 if in_norm is None else in_norm[2], 0,
                        0 if splitk_ws is None else splitk_ws.data_ptr(),
                        0 if splitk_ws is None else splitk_ws.numel() * splitk_ws.element_size(),
                        x.B, x.bs, out.bs, 0 if residual is None else residual.bs, 0,
                        0 if in_norm is None or len(in_norm) < 4 else in_norm[3],
                        0 if cw.w_wfrag is None else cw.w_wfrag.data_ptr())


class HipEngine:
    def __init__(self, state_dict, device, precision=None):
        self.lib = L.load()
        self.precision_name = precision or default_precision()
        if self.precision_name not in PRECISIONS:
            raise ValueError("otvm_amd: unknown precision %r (choose from %s)" % (self.precision_name, sorted(PRECISIONS)))
        self.precision = PRECISIONS[self.precision_name]
        self.conv_precision = CONV_PRECISIONS[self.precision_name]
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise RuntimeError("otvm_amd: the HIP path needs a GPU device (got %s); there is no CPU fallback" % device)
        self.sd = {k: v.detach().to(self.dev, torch.float32).contiguous() for k, v in state_dict.items()
                   if v.is_floating_point()}
        self.W = {}
        self._keep = []
        self.plans = {}
        self.bank = []
        self.free_slots = []
        self.stream = 0
        self.prof = None
        self.pending = None          # deferred memorize of the previous frame
        self.ev_dec = None           # fires when the last STM decoder has read the query encoder's buffers
        self.frame_counter = 0
        self.conv_calls = 0          # convolution launches issued by this engine (plan steps labelled "conv ", tuner launches included)
        self.last_T_read = 0
        self.parity = 0
        self.side = None
        self.side2 = None
        import os
        self.use_side_stream = os.environ.get("OTVM_SIDE_STREAM", "1") != "0"
        # f16x3 splits fp32 operands into fp16 halves: an activation beyond fp16's range (|x| >= 65504) becomes inf and
        # then NaN, silently, and would live on in the recurrent memory bank.  OTVM_CHECK_FINITE=1 checks every frame's
        # alpha (one device sync per frame) and raises; meant for the first run of a new checkpoint.
        # Round 3: the guard is ON by default in its cheap form (level 1): otvm_finite_guard scans what survives a frame --
        # the key / value maps entering the bank, the hidden state, the propagated trimap logits -- with |x| < 65504 and
        # records the first offending frame in a device flag; the host looks at the flag when it synchronises anyway
        # (last_frame, flush) and through a non-blocking copy every few frames.  Level 2 also synchronises and checks alpha
        # after every frame; level 3 additionally scans the input and output of every convolution (first run of a new
        # checkpoint); 0 switches everything off.
        self.check_level = int(os.environ.get("OTVM_CHECK_FINITE", "1") or 0)
        self.check_finite = self.check_level >= 2
        self.guard_flag = torch.full((1,), GUARD_CLEAR, dtype=torch.int32, device=self.dev)
        self._guard_host = torch.full((1,), GUARD_CLEAR, dtype=torch.int32).pin_memory()
        self._guard_ev = None
        self._guard_what = {}
        self.use_graphs = USE_GRAPHS
        self.gn_predict_off = False   # set by predict_check: the plans are (re)built without the predicted GroupNorm tails
        self.gn_predict_log = []      # (layer, kappa, action) of every intervention of the conditioning guard
        self.gn_predict_suspect_from = None   # frame of the current clip from which a mid-clip 'off' verdict makes alpha suspect
        self._diag_host = None
        self._diag_ev = None
        self.keep_hid_d = False      # training forward (otvm_amd/train.py): the decoder's hidden state must exist in memory
        global _TUNE_FILE_LOADED
        if not _TUNE_FILE_LOADED:
            _TUNE_FILE_LOADED = True
            with torch.cuda.device(self.dev):
                _load_tune_file()
        self._pack_all()

    # ------------------------------------------------------------------ weights
    def _stream(self):
        return torch.cuda.current_stream(self.dev).cuda_stream

    def _pack(self, name, w, ws=False, scale=None, bias=None, i_pad=None):
        cw = pack_conv_weight(self.lib, self.dev, w, ws, scale, i_pad, split=(self.precision == L.PREC_F16X3),
                              stream=self._stream())
        cw.bias = bias
        self._keep.append((w, scale))
        self.W[name] = cw

    def _wave_frag(self, cw):
        """MFMA B-fragment-major copy of a split weight (otvm_pack_wave_weight_f16x3), built once."""
        if cw.w_wfrag is None:
            O_pad = cw.w_hi.numel() // cw.K_pad
            cw.w_wfrag = torch.zeros(int(self.lib.otvm_wave_weight_bytes_f16x3(O_pad, cw.K_pad)), dtype=torch.uint8, device=self.dev)
            L.check(self.lib.otvm_pack_wave_weight_f16x3(cw.w_hi.data_ptr(), cw.w_lo.data_ptr(), O_pad, cw.K_pad,
                                                         cw.w_wfrag.data_ptr(), self._stream()), "pack_wave_weight")
        return cw.w_wfrag

    def _gram_tables(self, cw):
        return gram_tables(self.lib, cw)

    def _fold_bn(self, bn):
        sd = self.sd
        n = sd[bn + ".weight"].numel()
        scale = torch.empty(n, dtype=torch.float32, device=self.dev)
        bias = torch.empty(n, dtype=torch.float32, device=self.dev)
        L.check(self.lib.otvm_fold_bn(sd[bn + ".weight"].data_ptr(), sd[bn + ".bias"].data_ptr(),
                                      sd[bn + ".running_mean"].data_ptr(), sd[bn + ".running_var"].data_ptr(), 1e-5, n,
                                      scale.data_ptr(), bias.data_ptr(), self._stream()), "fold_bn " + bn)
        return scale, bias

    def _pack_all(self):
        with torch.cuda.device(self.dev):
            self._pack_all_on_device()

    def _pack_all_on_device(self):
        sd = self.sd
        for k, v in sd.items():
            if not (k.endswith(".weight") and v.dim() == 4):
                continue
            name = k[:-7]
            if k.startswith("NET."):
                ws = not ("conv_up4" in k or ".pred." in k)               # layers_WS.Conv2d vs nn.Conv2d
                if name in ("NET.decoder.conv_up4.4", "NET.refine.pred.4"):
                    continue                                             # 1x1 heads run inside otvm_fba_head
                # the two convs reading the 80-channel D80 buffer take all 80 channels (zero weights on the tail) so
                # that Cin is a multiple of 16 and the 3x3 patch kernel applies
                i_pad = 80 if name in ("NET.decoder.conv_up4.0", "NET.refine.conv1.0") else None
                self._pack(name, v, ws=ws, bias=sd.get(name + ".bias"), i_pad=i_pad)
            elif ".Encoder_" in k:
                if ".conv1_" in k or name.endswith("Encoder_M.conv1"):
                    continue                                             # merged stem, below
                bn = name.replace(".conv", ".bn").replace("downsample.0", "downsample.1")
                scale, bias = self._fold_bn(bn)
                self._pack(name, v, scale=scale, bias=bias)
            else:                                                        # KV heads, STM decoder: plain conv + bias
                self._pack(name, v, bias=sd.get(name + ".bias"))
        if FUSE_STM_BLOCK and self.precision == L.PREC_F16X3:
            for enc in ("trimap.model.Encoder_M.", "trimap.model.Encoder_Q."):
                if enc + "res2.0.conv3.weight" not in sd:
                    continue
                # first block: the projection shares the last GEMM -- one filter [scale3 W3 | scale_d Wd] over K = 128, split
                # with ONE power-of-two scale per output channel; the bias is the sum of the two folded biases
                s3, b3 = self._fold_bn(enc + "res2.0.bn3")
                sdn, bdn = self._fold_bn(enc + "res2.0.downsample.1")
                wcat = torch.cat([sd[enc + "res2.0.conv3.weight"] * s3[:, None, None, None],
                                  sd[enc + "res2.0.downsample.0.weight"] * sdn[:, None, None, None]], dim=1)
                self._pack(enc + "res2.0.conv3cat", wcat, bias=b3 + bdn)
                for b in range(3):
                    for cname in ("conv1", "conv2", "conv3cat" if b == 0 else "conv3"):
                        self._wave_frag(self.W[enc + "res2.%d.%s" % (b, cname)])
        # round 4: per bias-free 1x1 conv3 of the FBA bottlenecks the constants that turn its INPUT's channel sums and Gram
        # matrix into the GroupNorm sums of its output (csrc/gram.hip): v_g = sum of the group's filters, M_g = sum of their
        # outer products, the latter in the block-upper-triangular order of the Gram kernel's partials (off-diagonal x 2)
        self.GP = {}
        if FUSE_GN_PREDICT and self.precision == L.PREC_F16X3:
            for name, cw in list(self.W.items()):
                if (name.startswith("NET.encoder.layer") and name.endswith(".conv3") and cw.kh == 1 and cw.kw == 1 and cw.bias is None
                        and cw.I % 64 == 0 and cw.O % 32 == 0 and cw.I_pad == cw.I):
                    self.GP[name] = self._gram_tables(cw)
        # Encoder_M stem: conv1_h(hid16) + conv1(rgb) + conv1_m(p_un) + conv1_o(p_fg) + conv1_a(alpha) (STM.py:63-66)
        e = "trimap.model.Encoder_M."
        wcat = torch.cat([sd[e + "conv1_h.weight"], sd[e + "conv1.weight"], sd[e + "conv1_m.weight"],
                          sd[e + "conv1_o.weight"], sd[e + "conv1_a.weight"]], dim=1)
        scale, bias = self._fold_bn(e + "bn1")
        self._pack(e + "stem", wcat, scale=scale, bias=bias, i_pad=24)
        # conv_up1.0 = 3x3 over cat[layer4 (2048) | 4 PPM maps (4 x 256)] (FBA/models.py:362-365), weight-standardised over
        # the WHOLE filter: take the standardised weights from the packed copy and split them into the layer-4 part (a
        # convolution of its own) and the PPM part in the table layout otvm_ppm_conv_z reads ([scale][tap][c][o])
        up1 = self.W.get("NET.decoder.conv_up1.0")
        self.W_ppm = None
        if PPM_ALGEBRA and FUSE_PPM_HEAD and up1 is not None and (up1.O, up1.I, up1.kh, up1.kw) == (256, 3072, 3, 3):
            O_pad = up1.w.numel() // up1.K_pad
            wk = up1.w.view(O_pad, up1.K_pad)[:256, :9 * up1.I_pad].reshape(256, 3, 3, up1.I_pad)
            w_main = wk[..., :2048].permute(0, 3, 1, 2).contiguous()                       # OIHW, standardised already
            self._pack("NET.decoder.conv_up1.0.main", w_main, ws=False, bias=up1.bias)
            self.W_ppm = wk[..., 2048:3072].reshape(256, 9, 4, 256).permute(2, 1, 3, 0).contiguous()
        torch.cuda.synchronize(self.dev)

    # ------------------------------------------------------------------ range guard (f16x3 operands must stay in fp16 range)
    def guard(self, act, what, stream, tag=None):
        """Enqueue a scan of the NHWC view ``act`` (|x| < 65504, NaN fails) on ``stream``; level 0 = nothing."""
        if self.check_level <= 0:
            return
        tag = self.frame_counter - 1 if tag is None else tag
        # the exact-fp32 path has no range limit: only inf / NaN fail there
        limit = F16_LIMIT if self.precision == L.PREC_F16X3 else float("inf")
        for b in range(act.B):
            v = act.img(b) if act.B > 1 else act
            L.check(self.lib.otvm_finite_guard(v.ptr, v.P, v.C, v.ld, limit, max(int(tag), 0), self.guard_flag.data_ptr(), stream),
                    "finite_guard " + what)

    def scan_now(self, act, what, stream):
        """Level 3: scan ``act`` and look at the verdict immediately (one synchronisation per call)."""
        self.guard(act, what, stream)
        v = int(self.guard_flag.item())
        if v != GUARD_CLEAR:
            self.guard_flag.fill_(GUARD_CLEAR)
            raise FloatingPointError("otvm_amd: |x| >= 65504 (or inf / NaN) at the %s, frame %d" % (what, v))

    def _guard_raise(self, frame):
        self.guard_flag.fill_(GUARD_CLEAR)
        self._guard_host.fill_(GUARD_CLEAR)
        self._guard_ev = None
        raise FloatingPointError(
            "otvm_amd: a value outside fp16's range (|x| >= 65504, inf or NaN) reached the recurrent state (memorised key / "
            "value, hidden state or propagated trimap logits) at frame %d of the sequence: the f16x3 path splits fp32 operands "
            "into fp16 halves and cannot represent it.  Everything from that frame on is invalid.  Rerun with "
            "model.precision = 'f32' (exact-fp32 MFMA, no range limit), or OTVM_CHECK_FINITE=3 to find the layer." % frame)

    def guard_check(self, sync):
        """sync: read the flag now (one device synchronisation; the caller is about to synchronise anyway).  Otherwise look
        at the last non-blocking copy, if it has landed, and start a new one every 8 frames."""
        if self.check_level <= 0:
            return
        if sync:
            v = int(self.guard_flag.item())
            if v != GUARD_CLEAR:
                self._guard_raise(v)
            return
        if self._guard_ev is not None and self._guard_ev.query():
            self._guard_ev = None
            v = int(self._guard_host[0])
            if v != GUARD_CLEAR:
                self._guard_raise(v)
        if self._guard_ev is None and self.frame_counter % 8 == 0:
            self._guard_host.copy_(self.guard_flag, non_blocking=True)
            self._guard_ev = torch.cuda.Event()
            self._guard_ev.record(torch.cuda.current_stream(self.dev))

    # ------------------------------------------------------------------ conditioning guard of the predicted GroupNorm statistics
    def _predict_verdict(self, pl, words):
        """words: the plan's diagnostic words on the host.  Applies the policy above; returns "ok", "p3" (at least one layer was
        switched to the f16x3 Gram matrix) or "off"."""
        import struct
        import warnings
        verdict = "ok"
        for name, q, slot in pl._predicted:
            kappa = struct.unpack("f", struct.pack("i", int(words[2 * slot])))[0]
            flags = int(words[2 * slot + 1])
            if flags or kappa > GN_PREDICT_KAPPA_OFF:
                why = ("non-finite statistic" if flags & 1 else "fp16 operand saturated") if flags else "kappa %.3g" % kappa
                self.gn_predict_log.append((name, kappa, "off: " + why))
                if not self.gn_predict_off:
                    warnings.warn("otvm_amd: the predicted GroupNorm statistics of %s are ill-conditioned (%s): the prediction is "
                                  "switched off for this model (statistics accumulated from the convolution's output instead; "
                                  "OTVM_GN_PREDICT=0 does the same from the start)" % (name, why))
                self.gn_predict_off = True
                verdict = "off"
            elif kappa > GN_PREDICT_KAPPA_P3 and q.passes == 1:
                q.passes = 3                                   # (the launch list holds a reference to q: effective from the next launch)
                self.gn_predict_log.append((name, kappa, "f16x3 Gram matrix"))
                if verdict == "ok":
                    verdict = "p3"
        if verdict == "p3":
            # a captured hipGraph holds the Gram kernel chosen at CAPTURE time (1-pass or 3-pass): the lists are captured again
            # from their next use on.  The old graphs may still be executing: retired, not destroyed
            pl.retire_graphs()
        return verdict

    def predict_check(self, pl, sync):
        """sync: read the plan's diagnostic words now.  Otherwise look at the last non-blocking copy, if it has landed, and start a
        new one every 8 frames (as guard_check)."""
        if not pl._predicted or self.gn_predict_off:
            return "ok"
        if sync:
            words = pl.diag.cpu().tolist()
            pl.diag.zero_()
            self._diag_ev = None
            return self._predict_verdict(pl, words)
        verdict = "ok"
        if self._diag_ev is not None and self._diag_ev[0].query():
            ev, owner = self._diag_ev
            self._diag_ev = None
            if owner is pl:
                verdict = self._predict_verdict(pl, self._diag_host.tolist())
        if self._diag_ev is None and self.frame_counter % 8 == 0:
            if self._diag_host is None:
                self._diag_host = torch.zeros(2 * 64, dtype=torch.int32).pin_memory()
            self._diag_host.copy_(pl.diag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record(torch.cuda.current_stream(self.dev))
            self._diag_ev = (ev, pl)
        return verdict

    def _drop_plans(self):
        """Forget every plan and everything bound to one (bank slots carry launch parameters of the plan that made them)."""
        torch.cuda.synchronize(self.dev)
        self.plans.clear()
        self.bank, self.free_slots, self.pending = [], [], None
        self.ev_dec = None
        self.last_plan = None

    # ------------------------------------------------------------------ plans
    def plan(self, H, W, B=1):
        key = (H, W) if B == 1 else (H, W, B)
        if self.keep_hid_d:
            key = key + ("hid_d",)
        if key not in self.plans:
            self.plans[key] = FramePlan(self, H, W, B)
        return self.plans[key]

    def reset(self):
        self.free_slots.extend(self.bank)
        self.bank = []

    def _vec3(self, key):
        return [float(x) for x in self.sd[key].flatten().tolist()]

    def frame(self, *args, **kw):
        # the raw launches below go to torch's current stream of self.dev: make that device current for the call (a model
        # on cuda:N called without torch.cuda.set_device(N) would otherwise launch into another device's context)
        with torch.cuda.device(self.dev):
            return self._frame(*args, **kw)

    def frame_batch(self, *args, **kw):
        with torch.cuda.device(self.dev):
            return self._frame_batch(*args, **kw)

    def _frame(self, a, fg, bg, tri_gt=None, first_frame=False, last_frame=False, memorize=False, max_memory_num=2,
               dilate_kernel=None, frame_id=None, cls_override=None, frames_rgb=False, inputs_ready=None):
        """One call of EvalModel.forward (reference models/alpha/model.py:391-512) on the HIP path.

        a [1,1,1,H,W] in [0,1]; fg, bg [1,1,3,H,W] BGR 0..255; tri_gt [1,1,3,H,W] or None.
        Extension: fg / bg may be the decoded uint8 images themselves, [H,W,3] interleaved (BGR, or RGB with
        ``frames_rgb``): the preprocess kernel reads them directly (same arithmetic as ``.float()`` + permute).
        inputs_ready: True = a / fg / bg are device tensors whose contents are complete (nothing still in flight writes
        them); a torch.cuda.Event = complete once that event has fired; None = unknown (ordered after everything issued on
        the launch stream so far).  Only matters for how early the query encoder may start.
        Returns the reference's 5-tuple (scaled_imgs, preds_trimap, tri_gt, preds_alpha, scaled_gts)."""
        return self._frame_batch([a], [fg], [bg], [tri_gt], first_frame, last_frame, memorize, max_memory_num, dilate_kernel,
                                 frame_id, None if cls_override is None else [cls_override], frames_rgb, inputs_ready)[0]

    def _frame_batch(self, a_l, fg_l, bg_l, tri_gt_l, first_frame=False, last_frame=False, memorize=False, max_memory_num=2,
                     dilate_kernel=None, frame_id=None, cls_override=None, frames_rgb=False, inputs_ready=None, train=None,
                     _recheck=2):
        """The same frame step for B independent sequences in LOCK-STEP (round 3): lists of B inputs of one resolution, one
        memory schedule (first_frame / last_frame / memorize / max_memory_num apply to all), per-sequence banks.  Every
        (train: buffers of the training-mode forward, otvm_amd/train.py -- the heads' full outputs and the raw logits are
        written there as well, and frame 0 memorises the ground-truth trimap.)  Every
        layer is ONE launch over the B images (otvm_conv_params.batch, otvm_gn_*_b, ...): the weights are read once, and
        the small maps of the encoders -- too few tiles for 256 CUs from one image -- fill the chip.  Each image is computed
        exactly as a batch-1 call computes it (same tiles, same summation order).  Returns a list of B 5-tuples."""
        dev, lib = self.dev, self.lib
        f32 = torch.float32
        B = len(a_l)
        if not (len(fg_l) == len(bg_l) == len(tri_gt_l) == B) or B < 1:
            raise ValueError("otvm_amd: a / fg / bg / tri_gt lists must have the same length")
        main = torch.cuda.current_stream(dev)
        if isinstance(inputs_ready, torch.cuda.Event):
            main.wait_event(inputs_ready)                     # before anything below (a conversion, too) reads the inputs
        ins, H, W, converted = [], None, None, False
        u8 = fg_l[0].dtype == torch.uint8
        for a, fg, bg in zip(a_l, fg_l, bg_l):
            a_in, fg_in, bg_in = a, fg, bg
            a = a.to(dev, f32).contiguous()
            if (fg.dtype == torch.uint8) != u8:
                raise ValueError("otvm_amd: the frames of a batch must share one dtype")
            if u8:
                if bg.dtype != torch.uint8 or fg.dim() != 3 or fg.shape[-1] != 3 or bg.shape != fg.shape:
                    raise ValueError("otvm_amd: uint8 frames must be [H,W,3] for both fg and bg, got %s / %s"
                                     % (tuple(fg.shape), tuple(bg.shape)))
                fg, bg = fg.to(dev).contiguous(), bg.to(dev).contiguous()
                h_, w_ = int(fg.shape[0]), int(fg.shape[1])
            else:
                if frames_rgb:
                    raise ValueError("otvm_amd: frames_rgb applies to uint8 [H,W,3] frames only (fp32 input is BGR planar)")
                fg = fg.to(dev, f32).contiguous()
                bg = bg.to(dev, f32).contiguous()
                h_, w_ = int(fg.shape[-2]), int(fg.shape[-1])
            if H is None:
                H, W = h_, w_
            elif (h_, w_) != (H, W):
                raise ValueError("otvm_amd: the sequences of a batch must share one resolution (%dx%d vs %dx%d)" % (w_, h_, W, H))
            converted = converted or a is not a_in or fg is not fg_in or bg is not bg_in
            if a.dim() != 5 or a.shape[0] != 1 or a.shape[1] != 1 or tuple(a.shape[-2:]) != (H, W):
                raise ValueError("otvm_amd: inputs must be [1,1,C,H,W] (batch 1, one frame), got a %s for %dx%d frames"
                                 % (tuple(a.shape), H, W))
            ins.append((a, fg, bg))
        if inputs_ready is not None and converted:
            # an input was converted / copied just now BY THE LAUNCH STREAM (host tensor, other dtype, non-contiguous): the
            # caller's promise covers the tensor it passed, not this copy -- the side streams must order themselves behind
            # the launch stream (the conservative path), or the query encoder would read the copy before it is written
            inputs_ready = None
        pl = self.plan(H, W, B)
        stream = self._stream()
        outs = [dict(scaled_imgs=torch.empty((1, 1, 3, H, W), dtype=f32, device=dev),
                     alpha=torch.empty((1, 1, 1, H, W), dtype=f32, device=dev),
                     alpha_u8=torch.empty((H, W), dtype=torch.uint8, device=dev),
                     tri_out=torch.empty((1, 1, 3, H, W), dtype=f32, device=dev),
                     tri_gt_out=torch.empty((1, 1, 3, H, W), dtype=f32, device=dev)) for _ in range(B)]
        # ---- deferred memorize of the previous frame (reference order: memorize(t) ends frame t, alpha/model.py:461-493;
        # here it opens frame t+1 so that Encoder_M(t) and Encoder_Q(t+1), two chains of small launches that cannot fill
        # 256 CUs on their own, run concurrently on two HIP streams; they meet before the memory read)
        if first_frame:
            if self.pending is not None:                      # previous clip ended without last_frame=True: its deferred
                self.free_slots.append(self.pending["slot"])  # memorize is dropped, the slot goes back to the pool
            self.pending = None
            self.reset()
            self.frame_counter = 0
            self.gn_predict_suspect_from = None
        if frame_id is None:                                  # position in the sequence since the last first_frame
            frame_id = self.frame_counter
        self.frame_counter += 1
        self.guard_check(sync=False)
        pend, self.pending = self.pending, None
        if pend is not None and pend["plan"] is not pl:
            self._memorize(pend, stream)                      # resolution / batch changed: finish it in order
            pend = None
        par = self.parity
        self.parity ^= 1
        pps = []
        for (a, fg, bg) in ins:
            pp = L.PreprocessParams()
            pp.a = a.data_ptr()
            if u8:
                pp.fg_u8, pp.bg_u8, pp.u8_rgb = fg.data_ptr(), bg.data_ptr(), 1 if frames_rgb else 0
            else:
                pp.fg, pp.bg = fg.data_ptr(), bg.data_ptr()
            pp.H, pp.W, pp.Hp, pp.Wp, pp.lh, pp.lw = H, W, pl.Hp, pl.Wp, pl.lh, pl.lw
            for name, key in (("mean", "IMG_MEAN"), ("std", "IMG_STD"), ("mean_q", "trimap.model.Encoder_Q.mean"),
                              ("std_q", "trimap.model.Encoder_Q.std"), ("mean_m", "trimap.model.Encoder_M.mean"),
                              ("std_m", "trimap.model.Encoder_M.std")):
                setattr(pp, name, (C.c_float * 3)(*self._consts(key)))
            pps.append(pp)
        # ---- the query encoder (STM.segment's Encoder_Q + KV_Q) needs nothing but the frame itself.  It runs on a side
        # stream: next to Encoder_M of the previous frame (two chains of small launches that cannot fill 256 CUs on their
        # own), and -- when the caller vouches that the frame is already in device memory (inputs_ready) and the host runs
        # ahead of the device -- already under the previous frame's alpha network, which is still executing on the launch
        # stream.  Its buffers (SQ, the q_ trunk, QK, M4[512:]) were last read by the previous frame's STM decoder (ev_dec).
        use_side = self.prof is None and self.use_side_stream and not first_frame
        ev_q = ev_s2 = None
        split = None                                          # memory read started on a side stream (see below)
        if use_side:
            if self.side is None:
                self.side = torch.cuda.Stream(device=dev)
            side = self.side
            if self.ev_dec is not None:
                side.wait_event(self.ev_dec)
            if isinstance(inputs_ready, torch.cuda.Event):
                side.wait_event(inputs_ready)                # (the launch stream waited at the top of the call)
            elif inputs_ready is not True:                   # unknown producer: everything issued so far on the launch stream
                ev_in = torch.cuda.Event()
                ev_in.record(main)
                side.wait_event(ev_in)
            for b, (a, fg, bg) in enumerate(ins):
                for t_ in (a, fg, bg):                       # read on the side stream too: keep the allocator from reusing them early
                    t_.record_stream(side)
                pq = L.PreprocessParams.from_buffer_copy(pps[b])
                sq = pl.SQ.img(b)
                pq.sq, pq.sq_ld = sq.ptr, sq.ld
                L.check(lib.otvm_preprocess(C.byref(pq), side.cuda_stream), "preprocess (query encoder input)")
            pl.run("segment_a", side.cuda_stream, side)
            # The memory read is merged from per-chunk partials, so the bank may be visited in any grouping: every slot but
            # the one the previous frame is about to memorise is resident already -- its partials are computed here, on
            # the side stream, right behind the query key (i.e. also under the previous frame's alpha network); the launch
            # stream later adds the one new slot and merges (1/T of the read stays on the critical path).
            ev_q = torch.cuda.Event()
            ev_q.record(side)
            # ---- second side stream: the DENSE work of this frame that depends on the query encoder only -- the memory read
            # over the slots that are already resident (the read is merged from per-chunk partials, so the bank may be
            # visited in any grouping; only the slot the previous frame is about to memorise is missing) and the decoder's
            # skip branches -- is held back until the previous frame's alpha network has finished (ev_m0) and then runs
            # NEXT TO Encoder_M of the previous frame: that chain of small launches is on the critical path and cannot
            # fill 256 CUs, these kernels can.  The launch stream then adds the one new slot, merges, and runs the rest of
            # the decoder.
            if self.side2 is None:
                self.side2 = torch.cuda.Stream(device=dev)
            side2 = self.side2
            ev_m0 = torch.cuda.Event()
            ev_m0.record(main)
            side2.wait_event(ev_q)
            side2.wait_event(ev_m0)
            # the frame's own preprocess (composite, normalised copies for the alpha network and the next memorize) and the
            # clearing of the GroupNorm statistics need neither encoder: off the launch stream's serial chain
            if PRE_ON_S2:
                for b, (a, fg, bg) in enumerate(ins):
                    for t_ in (a, fg, bg):
                        t_.record_stream(side2)
                    outs[b]["scaled_imgs"].record_stream(side2)
                pl.clear_stats(side2.cuda_stream)
                for b in range(B):
                    self._preprocess_rest(pl, pps[b], par, outs[b]["scaled_imgs"], False, side2.cuda_stream, b)
            if self.precision == L.PREC_F16X3:
                if pend is not None:
                    nb, _ = bank_update(self.bank, pend["slot"], pend["first_frame"], pend["memorize"], pend["max_memory_num"])
                    old = [s_ for s_ in nb if s_ is not pend["slot"]]
                    fresh = [pend["slot"]] if any(s_ is pend["slot"] for s_ in nb) else []
                else:
                    old, fresh = list(self.bank), []
                if old:
                    split = pl.memory_read_begin(old, fresh, side2.cuda_stream)
            pl.run("segment_skip", side2.cuda_stream, side2)
            ev_s2 = torch.cuda.Event()
            ev_s2.record(side2)
        # ---- deferred memorize of the previous frame (reference order: memorize(t) ends frame t, alpha/model.py:461-493;
        # here it opens frame t+1 on the launch stream, concurrently with the query encoder above)
        if pend is not None:
            self._memorize(pend, stream)
        if not (use_side and PRE_ON_S2):
            pl.clear_stats(stream)
            for b in range(B):
                self._preprocess_rest(pl, pps[b], par, outs[b]["scaled_imgs"], not use_side, stream, b)

        tri_srcs = []
        for b in range(B):
            tri_gt, tri_gt_out, a = tri_gt_l[b], outs[b]["tri_gt_out"], ins[b][0]
            if tri_gt is not None:
                tri_src = tri_gt.to(dev, f32).contiguous()
                L.check(lib.otvm_onehot_argmax3(tri_src.data_ptr(), H * W, tri_gt_out.data_ptr(), stream), "onehot")
            else:
                if dilate_kernel is None:
                    raise ValueError("otvm_amd: tri_gt=None needs a fixed dilate_kernel (reference eval always sets one)")
                ws = pl.raws("tfa_ws", H * W, torch.uint8)[b]
                L.check(lib.otvm_trimap_from_alpha(a.data_ptr(), H, W, int(dilate_kernel), tri_gt_out.data_ptr(),
                                                   ws.data_ptr(), stream), "trimap_from_alpha")
                tri_src = tri_gt_out
            tri_srcs.append(tri_src)

        self.last_T_read = 0
        if first_frame:
            for b in range(B):
                L.check(lib.otvm_pad_trimap(tri_srcs[b].data_ptr(), H, W, pl.PROBS_B[b].data_ptr(), pl.Hp, pl.Wp, pl.lh, pl.lw,
                                            stream), "pad_trimap")
            # frame 1's query encoder (side stream) is ordered behind everything the launch stream was handed up to here:
            # loop-invariant inputs the caller created for this clip (the trimap flow's constant alpha) are complete
            self.ev_dec = torch.cuda.Event()
            self.ev_dec.record(main)
        else:
            if ev_q is not None:
                main.wait_event(ev_q)
            else:
                pl.run("segment_a", stream)
                pl.run("segment_skip", stream)
            if not self.bank:
                raise RuntimeError("otvm_amd: non-first frame with an empty memory bank (call with first_frame=True first)")
            self.last_T_read = len(self.bank)
            if self.prof is not None:                         # bench.py roofline leg: HIP events around the memory read
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            if split is not None:
                if len(split["old"]) + len(split["fresh"]) != len(self.bank) or not all(
                        any(b_ is s_ for b_ in self.bank) for s_ in split["old"] + split["fresh"]):
                    raise RuntimeError("otvm_amd: the early memory read visited a different bank than the policy produced")
                pl.memory_read_fresh(split, stream)           # the slot just memorised: needs only the query key (ev_q)
                main.wait_event(ev_s2)                        # partials of the resident slots + the decoder's skip branches
                pl.memory_read_merge(split, stream)
            else:
                if ev_s2 is not None:
                    main.wait_event(ev_s2)
                pl.memory_read(self.bank, stream)
            if self.prof is not None:
                e1.record()
                self.prof.append(("memory_read", 1280.0 * len(self.bank) * pl.hw * pl.hw * B, e0, e1, len(self.bank)))
            pl.run("segment_b", stream)
            self.guard(pl.L4, "propagated trimap logits", stream, frame_id)
            if train is not None and train.get("seg_logits") is not None:
                for b in range(B):                             # the cross-entropy takes the logits (model.py:286-288)
                    l4 = pl.L4.img(b)
                    L.check(lib.otvm_upsample4_logits3(l4.ptr, l4.H, l4.W, l4.ld, train["seg_logits"][b].data_ptr(), stream),
                            "upsample4_logits3")
            if not EVDEC_LATE:
                self.ev_dec = torch.cuda.Event()              # the query encoder's buffers are free again
                self.ev_dec.record(main)
        pl.encode(stream, cls_override)
        if not first_frame and EVDEC_LATE:
            # round 4: the NEXT frame's query encoder (side stream) is released behind the trimap encoding, not in front of
            # it: its fused-bottleneck kernels (one workgroup per CU, every register of the CU) otherwise take the chip from
            # the three small kernels of the encoding -- on the frame's serial chain -- which then wait for whole workgroups
            # to drain (edt_rows_encode 204 us inside the frame against 90 us alone); the encoder still has the whole alpha
            # network to hide under
            self.ev_dec = torch.cuda.Event()
            self.ev_dec.record(main)
        pl.run("fba", stream)
        pl.run("fba_tail%d" % par, stream)
        self.guard(pl.SMs[par].ch(0, 16), "hidden state", stream, frame_id)
        if train is not None:
            # training forward (model.py:226-233,259-275): the fused (alpha, F, B) of both heads and the refinement's trimap
            # logits feed the losses; frame 0 memorises the ground-truth trimap (preds_trimap_refine[0] = tri[:, 0])
            sd, P_ = self.sd, pl.P
            img = pl.D80.ch(67, 3)
            for b in range(B):
                hd, hr = pl.HID_D.img(b), pl.SMs[par].ch(0, 16).img(b)
                L.check(lib.otvm_fba_head_train(hd.ptr, hd.ld, sd["NET.decoder.conv_up4.4.weight"].data_ptr(),
                                                sd["NET.decoder.conv_up4.4.bias"].data_ptr(), 7, img.img(b).ptr, img.ld, P_,
                                                train["dec7"][b].data_ptr(), 0, stream), "fba_head_train (decoder)")
                L.check(lib.otvm_fba_head_train(hr.ptr, hr.ld, sd["NET.refine.pred.4.weight"].data_ptr(),
                                                sd["NET.refine.pred.4.bias"].data_ptr(), 10, img.img(b).ptr, img.ld, P_,
                                                train["ref7"][b].data_ptr(), train["ref_logits"][b].data_ptr(), stream),
                        "fba_head_train (refinement)")
                if train.get("tri0") is not None:
                    smv = pl.SMs[par].ch(16, 8).img(b)
                    L.check(lib.otvm_trimap_to_sm(train["tri0"][b].data_ptr(), P_, smv.ptr, smv.ld, stream), "trimap_to_sm")
        if not last_frame:
            slot = pl.new_slot()
            slot["frame"] = frame_id
            self.pending = dict(plan=pl, par=par, slot=slot, first_frame=first_frame, memorize=memorize,
                                max_memory_num=max_memory_num)
        for b in range(B):
            o = outs[b]
            L.check(lib.otvm_crop_outputs(pl.ALPHA_P_B[b].data_ptr(), pl.TRI_P_B[b].data_ptr(), pl.Hp, pl.Wp, H, W, pl.lh, pl.lw,
                                          o["alpha"].data_ptr(), o["alpha_u8"].data_ptr(), o["tri_out"].data_ptr(), stream), "crop")
        self.last_alpha_u8 = outs[0]["alpha_u8"]
        self.last_alpha_u8_b = [o["alpha_u8"] for o in outs]
        self.last_plan = pl
        if pl._predicted and not self.gn_predict_off:
            # conditioning guard of the predicted GroupNorm statistics (see GN_PREDICT_KAPPA_*): a clip's first frame is checked
            # before it is returned (one synchronisation per clip) and computed again if a layer had to change its route -- the
            # bank is empty there, so the frame call is repeatable; later frames are watched without synchronising
            verdict = self.predict_check(pl, sync=first_frame)
            if verdict == "off" and not first_frame:
                # detected in the middle of a clip (asynchronous words, up to two check periods old): the plans are rebuilt at the
                # next first_frame; the frames of THIS clip from `suspect` on were normalised with statistics now known to be
                # ill-conditioned -- flagged for the caller (gn_predict_suspect_from, reset at the next first_frame), not hidden
                import warnings
                self.gn_predict_suspect_from = max(0, self.frame_counter - 1 - 16)
                warnings.warn("otvm_amd: predicted GroupNorm statistics went ill-conditioned in the middle of a clip: its frames "
                              "from %d on are suspect (engine.gn_predict_suspect_from); the rest of the clip keeps the current "
                              "plan, the next clip runs on the accumulated route" % self.gn_predict_suspect_from)
            if first_frame and verdict != "ok" and _recheck > 0:
                if verdict == "off":
                    self._drop_plans()                        # (rebuilt below without the predicted tails)
                return self._frame_batch(a_l, fg_l, bg_l, tri_gt_l, first_frame, last_frame, memorize, max_memory_num, dilate_kernel,
                                         frame_id, cls_override, frames_rgb, None, train, _recheck=_recheck - 1)
        elif self.gn_predict_off and pl._predicted and first_frame:
            # the guard tripped in the middle of the previous clip: this clip's first frame ran on a plan that still predicts --
            # rebuild now and compute it again
            self._drop_plans()
            return self._frame_batch(a_l, fg_l, bg_l, tri_gt_l, first_frame, last_frame, memorize, max_memory_num, dilate_kernel,
                                     frame_id, cls_override, frames_rgb, None, train, _recheck=0)
        if self.check_finite and not all(bool(torch.isfinite(o["alpha"]).all()) for o in outs):
            raise FloatingPointError("otvm_amd: non-finite alpha at frame %d -- an activation probably left fp16's range on the "
                                     "f16x3 path; rerun with model.precision = 'f32' (exact-fp32 MFMA)" % frame_id)
        if last_frame or self.check_finite:
            # end of the clip: the caller synchronises to collect its results -- the guard's verdict for the whole clip is
            # read here (the reference's loop synchronises after every frame, eval.py:195-197)
            self.guard_check(sync=True)
        return [(o["scaled_imgs"], o["tri_out"], o["tri_gt_out"], o["alpha"], ins[b][0]) for b, o in enumerate(outs)]

    def _preprocess_rest(self, pl, pp, par, scaled_imgs, with_sq, stream, b=0):
        """otvm_preprocess of image ``b`` for everything but (with_sq False) the query encoder's input: the composite returned
        to the caller, the alpha network's input channels, the next memorize's image channels, the refinement's image channels."""
        pm = L.PreprocessParams.from_buffer_copy(pp)
        pm.scaled_imgs = scaled_imgs.data_ptr()
        x11 = pl.X11.img(b)
        pm.x11, pm.x11_ld = x11.ptr, x11.ld
        if with_sq:
            sq = pl.SQ.img(b)
            pm.sq, pm.sq_ld = sq.ptr, sq.ld
        smv = pl.SMs[par].ch(16, 8).img(b)
        pm.sm, pm.sm_ld = smv.ptr, smv.ld
        d80 = pl.D80.img(b)
        pm.d80, pm.d80_ld = d80.ptr, d80.ld
        L.check(self.lib.otvm_preprocess(C.byref(pm), stream), "preprocess")

    def _memorize(self, pend, stream, tstream=None):
        """STM.memorize of a finished frame + the bank policy (alpha/model.py:466-493), on ``stream``."""
        pl, slot = pend["plan"], pend["slot"]
        pl.run("mem_stem%d" % pend["par"], stream, tstream)
        pl.run("mem_trunk", stream, tstream)
        pl.kv_into_slot(slot, stream)
        self.bank, released = bank_update(self.bank, slot, pend["first_frame"], pend["memorize"], pend["max_memory_num"])
        self.free_slots.extend(released)

    def bank_frames(self):
        """Frame ids resident in the bank, including a memorize that is still deferred (host-side policy only)."""
        bank = list(self.bank)
        pend = self.pending
        if pend is not None:
            bank, _ = bank_update(bank, pend["slot"], pend["first_frame"], pend["memorize"], pend["max_memory_num"])
        return [s["frame"] for s in bank]

    def flush(self):
        """Run a deferred memorize now (bank introspection, end of stream)."""
        pend, self.pending = self.pending, None
        if pend is not None:
            with torch.cuda.device(self.dev):
                self._memorize(pend, self._stream())
                self.guard_check(sync=True)

    def _consts(self, key):
        c = getattr(self, "_const_cache", None)
        if c is None:
            c = self._const_cache = {}
        if key not in c:
            c[key] = self._vec3(key)
        return c[key]


class FramePlan:
    """Buffers + launch lists for one input resolution."""

    def __init__(self, eng, H, W, B=1):
        self.e = eng
        self.lib = eng.lib
        self.dev = eng.dev
        self.H, self.W, self.B = H, W, B
        if B > 1 and eng.precision != L.PREC_F16X3:
            raise NotImplementedError("otvm_amd: batched sequences (B > 1) run on the f16x3 path only")
        self.lw, self.uw, self.lh, self.uh = pad_amounts(H, W, 32)
        self.Hp, self.Wp = H + self.lh + self.uh, W + self.lw + self.uw
        self.P = self.Hp * self.Wp
        self._bufs = {}
        self._keep = []
        self.graphs, self._graph_warm = {}, {}
        self._n_conv = {}
        self._retired_graphs = []
        self._stm128 = []                                     # launch lists holding an unresolved planes-128 STM block
        self._fused_stats = []
        self._convs = []
        self._predicted = []                                  # (layer, otvm_gram_params, slot in self.diag) of every predicted tail
        self.diag = None
        self.n_gn = 0
        self.steps = {}
        self._build()
        self.stats_bs = max(self.n_gn, 1) * 64                # doubles per image: one [32][2] block per GroupNorm
        self.stats = torch.zeros(self.B * self.stats_bs, dtype=torch.float64, device=self.dev)
        self._bind_stats()
        self.autotune()

    # ---- plan-time autotuning (see AUTOTUNE above)
    @staticmethod
    def _signature(p):
        return (p.H, p.W, p.Cin, p.in_ld, p.Cout, p.out_ld, p.kh, p.kw, p.stride, p.pad, p.dil, p.in_relu, p.act, int(bool(p.bias)),
                int(bool(p.residual)), p.res_ld, int(bool(p.gn_stats)), int(bool(p.in_scale)), int(bool(p.splitk_ws)), p.precision,
                max(1, p.batch), int(bool(p.res_scale)), int(bool(p.in_res)))

    def _time_conv(self, p, code, stream, reps=3):
        p.tune = code
        fn = self.lib.otvm_conv2d
        self.e.conv_calls += 1 + reps
        L.check(fn(C.byref(p), stream), "autotune warm-up")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn(C.byref(p), stream)
        e1.record()
        return e0, e1, reps

    def tune_convs(self, convs):
        """Pick otvm_conv_params.tune for every (params, name) in ``convs`` (cached per layer signature)."""
        if not (AUTOTUNE and self.e.precision == L.PREC_F16X3):
            return
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        codes = (C.c_int * 128)()
        timed_any = False
        for p, name in convs:
            sig = self._signature(p)
            if sig not in _TUNE_CACHE:
                n = int(self.lib.otvm_conv2d_candidates(C.byref(p), codes, 128))
                cands = [0] + [int(codes[i]) for i in range(n)]           # 0 = the built-in heuristic, the incumbent
                if not WAVE_TILE:                                         # (the fragment-major weights alone do not switch the one-wave tile on)
                    cands = [c for c in cands if c // 16 - 1 != 9]
                # the first timing of a layer runs on cold caches and ramping clocks (measured: the same kernel 14 % slower
                # as first candidate than as second): a throw-away pass first, the incumbent timed again at the end
                self._time_conv(p, 0, stream, reps=2)
                timed = [(c,) + self._time_conv(p, c, stream) for c in cands + [0]]
                torch.cuda.synchronize(self.dev)
                ms = {}
                for c, e0, e1, r in timed:
                    t = e0.elapsed_time(e1) / r
                    ms[c] = min(ms.get(c, t), t)
                best = min(ms, key=ms.get)
                if ms[best] > 0.97 * ms[0]:                                # keep the incumbent unless clearly beaten
                    best = 0
                _TUNE_CACHE[sig] = best
                TUNE_LOG.append((name, sig, best, ms))
                timed_any = True
            p.tune = _TUNE_CACHE[sig]
        if timed_any:
            _save_tune_file()

    def _resolve_stm128(self, timed):
        """Every planes-128 STM identity block of the plan: fused (one launch, pixel tile 1 = 8x16 / 2 = 8x8 / 3 = 4x8) or its three
        convolution launches (0), whichever is faster on this map -- timed once per (map, batch) with the buffers holding random
        values (autotune), cached beside the conv tuner's choices; FUSE_STM_BLOCK128 = 2 or no tuner: fused, tile from the map."""
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        timed_any = False
        for S in self._stm128:
            i = 0
            while i < len(S):
                st = S[i]
                if not (isinstance(st[0], str) and st[0] == "stm128"):
                    i += 1
                    continue
                _, fused, U, q, name = st
                key = (-128, q.H, q.W, max(1, q.batch), q.x_ld, q.y_ld)
                if FUSE_STM_BLOCK128 >= 2 or not timed:
                    choice = _TUNE_CACHE.get(key, -1) if FUSE_STM_BLOCK128 < 2 else (STM128_TILE or -1)
                elif key in _TUNE_CACHE:
                    choice = _TUNE_CACHE[key]
                else:
                    def run(c, reps):
                        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                        e0.record()
                        for _ in range(reps):
                            if c == 0:
                                for u in U:
                                    u[0](*u[1], stream)
                            else:
                                q.tile = c
                                L.check(fused[0](*fused[1], stream), name)
                        e1.record()
                        return e0, e1, reps
                    self.e.conv_calls += 3 * 7 + 3 * 7                  # (counted like the conv tuner's launches)
                    run(0, 2)
                    timed_ = [(c,) + run(c, 3) for c in (0, 1, 2, 3, 0, 1, 2, 3)]
                    torch.cuda.synchronize(self.dev)
                    ms = {}
                    for c, e0, e1, r in timed_:
                        t = e0.elapsed_time(e1) / r
                        ms[c] = min(ms.get(c, t), t)
                    choice = min(ms, key=ms.get)
                    _TUNE_CACHE[key] = choice
                    TUNE_LOG.append((name + " (fused bottleneck: 0 = three launches, 1 / 2 / 3 = fused on 8x16 / 8x8 / 4x8 pixels)",
                                     (q.H, q.W, 512, q.x_ld, 512, q.y_ld, 3, 3, 1, 1, 1), choice, ms))
                    timed_any = True
                if choice == 0:
                    S[i:i + 1] = U
                    i += len(U)
                else:
                    q.tile = max(choice, 0)                              # (-1: the library picks from the map size)
                    S[i] = fused
                    i += 1
        self._stm128 = []
        if timed_any:
            _save_tune_file()

    def autotune(self):
        if not (AUTOTUNE and self.e.precision == L.PREC_F16X3):
            self._resolve_stm128(timed=False)
            return
        # realistic operand values while timing (zero-filled operands run the matrix cores at an unrepresentative power)
        filled = []
        gen = torch.Generator(device=self.dev)                 # a private generator: the caller's global RNG stream is not consumed
        gen.manual_seed(0x07F3)
        for b in self._bufs.values():
            t = b.t if isinstance(b, Act) else b
            if not t.is_floating_point():
                continue
            if t.dtype == torch.float32:
                t.normal_(generator=gen)
            else:
                continue
            filled.append(t)
        # the key / value convolutions of STM.memorize are bound to a bank slot at run time (kv_into_slot): time their
        # shapes here against scratch outputs, so the first memorised frame finds them in the cache
        H16, W16 = self.Hp // 16, self.Wp // 16
        tk = Act(torch.empty(self.B * self.hw * 128, dtype=torch.float32, device=self.dev), H16, W16, 128, B=self.B, bs=self.hw * 128)
        tv = Act(torch.empty(self.B * self.hw * 512, dtype=torch.float32, device=self.dev), H16, W16, 512, B=self.B, bs=self.hw * 512)
        n0, k0 = len(self._convs), len(self._keep)
        self.conv([], self.r4m, "trimap.model.KV_M_r4.Key", tk, pad=1)
        self.conv([], self.r4m, "trimap.model.KV_M_r4.Value", tv, pad=1)
        self.tune_convs(self._convs)
        self._resolve_stm128(timed=FUSE_STM_BLOCK128 == 1)
        torch.cuda.synchronize(self.dev)
        del self._convs[n0:], self._keep[k0:]                  # scratch-bound parameter blocks: not kept
        for t in filled:
            t.zero_()
        self.stats.zero_()
        torch.cuda.synchronize(self.dev)

    def retire_graphs(self):
        """Forget the captured hipGraphs (a launch parameter changed): every list runs directly once more and is captured again.
        The retired graphs stay alive with the plan -- one of them may still be executing."""
        self._retired_graphs.extend(self.graphs.values())
        self.graphs, self._graph_warm = {}, {}

    def clear_stats(self, stream):
        """Zero the GroupNorm statistics arena for the coming frame (library call on ``stream``, no ATen launch)."""
        L.check(self.lib.otvm_clear(self.stats.data_ptr(), self.stats.numel() * 8, stream), "clear GroupNorm statistics")

    # ---- buffers
    def buf(self, name, H, W, C):
        key = (name, H, W, C)
        if key not in self._bufs:
            self._bufs[key] = Act(torch.zeros(self.B * H * W * C, dtype=torch.float32, device=self.dev), H, W, C, B=self.B,
                                  bs=H * W * C)
        return self._bufs[key]

    def raw(self, name, n, dtype=torch.float32):
        key = ("raw", name, n, dtype)
        if key not in self._bufs:
            self._bufs[key] = torch.zeros(n, dtype=dtype, device=self.dev)
        return self._bufs[key]

    def raws(self, name, n, dtype=torch.float32):
        """One flat buffer of ``n`` elements PER IMAGE (planar tensors and workspaces of the per-image glue kernels)."""
        t = self.raw(name, n * self.B, dtype)
        return [t[b * n:(b + 1) * n] for b in range(self.B)]

    # ---- step builders (S = list being filled)
    def conv(self, S, x, wname, out, stride=1, pad=0, dil=1, act=NONE, in_relu=0, residual=None, in_norm=None, in_res=None):
        """in_res (with in_norm): x' = in_act(x * scale + shift + in_res) -- the identity of a residual block whose last apply
        pass is skipped (otvm_conv_params.in_res; 3x3 patch-kernel layers only)."""
        w = self.e.W[wname]
        assert x.C == w.I_pad, (wname, x.C, w.I_pad)
        Ho = (x.H + 2 * pad - dil * (w.kh - 1) - 1) // stride + 1
        Wo = (x.W + 2 * pad - dil * (w.kw - 1) - 1) // stride + 1
        assert (out.H, out.W) == (Ho, Wo) and out.C >= w.O, (wname, out.H, out.W, Ho, Wo, out.C, w.O)
        p = conv_params(x, w, out, w.bias, stride, pad, dil, act, in_relu, residual, self.e.conv_precision, in_norm, self._ws)
        if in_res is not None:
            assert in_norm is not None and (in_res.H, in_res.W) == (x.H, x.W) and in_res.C >= w.I
            p.in_res, p.in_res_ld, p.in_res_bs = in_res.ptr, in_res.ld, in_res.bs
        self._keep.append(p)
        self._convs.append((p, wname))
        flops = 2 * Ho * Wo * w.O * w.kh * w.kw * w.I * x.B     # algorithmic (un-padded) 2*MAC, all images of the launch
        # algorithmic bytes: read the input once, the weights once, write the output once (+ residual read), fp32
        abytes = 4 * (x.B * x.H * x.W * w.I * (2 if in_res is not None else 1) + w.O * w.I * w.kh * w.kw
                      + x.B * Ho * Wo * w.O * (2 if residual is not None else 1))
        S.append((self.lib.otvm_conv2d, (C.byref(p),), "conv " + wname, flops, abytes, (x, out.ch(0, _rup(w.O, 4)) if out.C >= _rup(w.O, 4) else out)))
        return p

    def conv_head(self, S, x, wname, hid_out, head_w, head_b, n_out, img, alpha, alpha_stride, alpha_bs, tri=None, sm=None):
        """3x3 conv 32 -> 16 + LeakyReLU with the FBA head in its epilogue (otvm_conv2d_head).  hid_out: Act or None (hidden
        state not written); img: Act view of the 3 image channels; alpha: (tensor pointer of image 0); tri: pointer or None;
        sm: Act view (channels 16..23 of the Encoder_M input) or None."""
        w = self.e.W[wname]
        assert x.C == w.I_pad and w.O == 16 and w.kh == 3
        out = hid_out if hid_out is not None else x
        p = conv_params(x, w, out, w.bias, 1, 1, 1, LEAKY, 0, None, self.e.conv_precision, None, None)
        if hid_out is None:
            p.out, p.out_ld, p.out_bs = 0, 0, 0
        h = L.HeadParams()
        h.w, h.b, h.n_out = head_w.data_ptr(), head_b.data_ptr(), n_out
        h.img, h.img_ld, h.img_bs = img.ptr, img.ld, img.bs
        h.P = x.P
        h.alpha_out, h.alpha_stride, h.alpha_bs = alpha, alpha_stride, alpha_bs
        h.tri_out, h.tri_bs = (0 if tri is None else tri), 3 * x.P
        if sm is not None:
            h.sm, h.sm_ld, h.sm_bs = sm.ptr, sm.ld, sm.bs
        if w.w16 is not None and HEAD16:
            h.w16 = w.w16.data_ptr()
        self._keep += [p, h]
        flops = 2 * x.P * w.O * 9 * w.I * x.B
        abytes = 4 * (x.B * x.P * w.I + w.O * w.I * 9 + x.B * x.P * (16 if hid_out is not None else 0) + x.B * x.P * 4)
        S.append((self.lib.otvm_conv2d_head, (C.byref(p), C.byref(h)), "conv " + wname + " (+ head)", flops, abytes,
                  (x, hid_out if hid_out is not None else x)))
        return p

    def gn(self, S, x, name, act, out=None, residual=None, conv_p=None, res_norm=None):
        """GroupNorm(32) of the raw conv output ``x``.  With ``conv_p`` (the params of the conv that produced x)
        the statistics are accumulated in that conv's epilogue and the separate stats pass is dropped.
        res_norm = (scale_ptr, shift_ptr, act): the residual is a raw GroupNorm input normalised on the fly."""
        out = x if out is None else out
        sd = self.e.sd
        idx = self.n_gn
        self.n_gn += 1
        if conv_p is not None and FUSE_GN_STATS:
            self._fused_stats.append((conv_p, idx))
        else:
            S.append(("gn_stats", (x.ptr, x.P, x.C, x.ld), idx, (self.B, x.bs), "gn_stats " + name))
        q = L.GnApplyParams()
        q.x, q.P, q.C, q.ld = x.ptr, x.P, x.C, x.ld
        q.gamma, q.beta = sd[name + ".weight"].data_ptr(), sd[name + ".bias"].data_ptr()
        if residual is not None:
            q.residual, q.res_ld, q.res_bs = residual.ptr, residual.ld, residual.bs
        if res_norm is not None:
            q.res_scale, q.res_shift, q.res_act = res_norm[0], res_norm[1], res_norm[2]
            q.norm_bs = res_norm[3] if len(res_norm) > 3 else 0
        q.act, q.out, q.out_ld = act, out.ptr, out.ld
        q.batch, q.x_bs, q.out_bs = self.B, x.bs, out.bs
        self._keep.append(q)
        S.append(("gn_apply", q, idx, "gn_apply " + name))

    def gn_table_step(self, S, x, gn_name, producer_p):
        """Per-channel scale / shift of GroupNorm(gn_name) over the raw conv output ``x`` (statistics from the producing
        conv's epilogue) for consumers that normalise on the fly instead of reading a normalised copy."""
        sd = self.e.sd
        idx = self.n_gn
        self.n_gn += 1
        self._fused_stats.append((producer_p, idx))
        tab = self.raw("gntab_" + gn_name, 2 * x.C * self.B)             # [B][scale C | shift C]
        if FUSE_GN_TABLE and isinstance(producer_p, L.ConvParams):
            # round 3 (ABI 16): the producing conv's last workgroup writes the table -- no launch between producer and consumer
            cnt = self.raw("gncnt_" + gn_name, self.B, torch.int32)
            producer_p.gn_gamma, producer_p.gn_beta = sd[gn_name + ".weight"].data_ptr(), sd[gn_name + ".bias"].data_ptr()
            producer_p.gn_scale_out, producer_p.gn_shift_out = tab.data_ptr(), tab.data_ptr() + 4 * x.C
            producer_p.gn_counter, producer_p.gn_tab_bs = cnt.data_ptr(), 2 * x.C
        else:
            S.append(("gn_table", (x.P, x.C, sd[gn_name + ".weight"].data_ptr(), sd[gn_name + ".bias"].data_ptr(),
                                   tab.data_ptr(), tab.data_ptr() + 4 * x.C), idx, 2 * x.C, "gn_table " + gn_name))
        return tab.data_ptr(), tab.data_ptr() + 4 * x.C, 2 * x.C

    def gn_then_conv(self, S, x, gn_name, gn_act, producer_p, wname, out, **kw):
        """GroupNorm(32) + activation of the raw conv output ``x`` whose ONLY consumer is the conv ``wname``.  When
        that conv runs on the patch kernel the apply is folded into its input staging (x stays raw in memory, the
        normalised tensor is never written); otherwise the usual in-place apply pass is emitted."""
        w = self.e.W[wname]
        sd = self.e.sd
        if FUSE_GN_APPLY and FUSE_GN_STATS and producer_p is not None:
            probe = conv_params(x, w, out, w.bias, kw.get("stride", 1), kw.get("pad", 0), kw.get("dil", 1), kw.get("act", NONE),
                                0, kw.get("residual"), self.e.conv_precision, (1, 1, gn_act))
            kind = self.lib.otvm_conv2d_input_norm_kind(C.byref(probe))
            # the implicit-GEMM tiles normalise every input element once per TAP: a win for 1x1 layers (bn2 -> conv3), a loss
            # on the 3x3 layers the patch kernel does not take (layer 3 / 4 of the FBA encoder: 0.145 vs 0.126 ms and 0.434 vs
            # 0.394 ms per launch against a 13-20 us apply pass) -- those keep their pass
            if kind == 1 or (kind == 2 and (w.kh * w.kw == 1 or FUSE_GN_APPLY_IGEMM_KXK)):
                sc, sh, nbs = self.gn_table_step(S, x, gn_name, producer_p)
                return self.conv(S, x, wname, out, in_norm=(sc, sh, gn_act, nbs), **kw)
        self.gn(S, x, gn_name, gn_act, conv_p=producer_p)
        return self.conv(S, x, wname, out, **kw)

    def _bind_stats(self):
        base, sbs = self.stats.data_ptr(), self.stats_bs
        for conv_p, idx in self._fused_stats:
            conv_p.gn_stats = base + idx * 512
            conv_p.gn_bs = sbs
        for key, S in self.steps.items():
            for i, st in enumerate(S):
                if st[0] == "gn_stats":
                    _, a, idx, (nb, x_bs), label = st
                    S[i] = (self.lib.otvm_gn_stats_b, a + (base + idx * 512, nb, x_bs, sbs), label)
                elif st[0] == "gn_table":
                    _, a, idx, nbs, label = st
                    S[i] = (self.lib.otvm_gn_table_b, (base + idx * 512,) + a + (self.B, sbs, nbs), label)
                elif st[0] == "ppm_add":
                    _, a, b, label = st
                    S[i] = (self.lib.otvm_ppm_conv_add, a + (self._ppm_stats.gn_stats + 8 * b * sbs,), label)
                elif st[0] == "gn_apply":
                    _, q, idx, label = st
                    q.stats, q.stats_bs = base + idx * 512, sbs
                    S[i] = (self.lib.otvm_gn_apply_b, (C.byref(q),), label)

    def upsample(self, S, x, out, add=None, norm=None):
        """norm = (scale_ptr, shift_ptr, act, table stride): x is a raw GroupNorm input, normalised per source pixel while
        resampling."""
        S.append((self.lib.otvm_upsample_bilinear_b,
                  (x.ptr, x.H, x.W, x.C, x.ld, 0 if norm is None else norm[0], 0 if norm is None else norm[1],
                   0 if norm is None else norm[2], 0 if add is None else add.ptr, 0 if add is None else add.ld,
                   out.ptr, out.H, out.W, out.ld, x.B, x.bs, 0 if add is None else add.bs, out.bs,
                   0 if norm is None else norm[3]), "upsample"))

    def gn_then_upsample(self, S, x, gn_name, gn_act, producer_p, out):
        """GroupNorm + activation of the raw conv output ``x`` whose only consumer is a bilinear resampling: the apply
        pass is folded into the resampling kernel (x stays raw, the normalised tensor is never written)."""
        if FUSE_GN_APPLY and FUSE_GN_STATS and producer_p is not None:
            sc, sh, nbs = self.gn_table_step(S, x, gn_name, producer_p)
            self.upsample(S, x, out, norm=(sc, sh, gn_act, nbs))
        else:
            self.gn(S, x, gn_name, gn_act, conv_p=producer_p)
            self.upsample(S, x, out)

    def maxpool(self, S, x, out):
        S.append((self.lib.otvm_maxpool3x3s2_b, (x.ptr, x.H, x.W, x.C, x.ld, out.ptr, out.ld, x.B, x.bs, out.bs), "maxpool"))

    # ---- network pieces
    def gn_bottleneck(self, S, x, p, planes, stride, dil, has_ds, out):
        Ho, Wo = x.H // stride, x.W // stride
        t1 = self.buf("bt1", x.H, x.W, planes)
        cp = self.conv(S, x, p + ".conv1", t1)
        t2 = self.buf("bt2", Ho, Wo, planes)
        cp = self.gn_then_conv(S, t1, p + ".bn1", RELU, cp, p + ".conv2", t2, stride=stride, pad=dil, dil=dil)
        if self._predicted_tail(S, x, p, planes, stride, has_ds, out, t2, cp):
            return
        t3 = self.buf("bt3", Ho, Wo, planes * 4)
        if FUSE_GN_APPLY_IGEMM:
            # round 3: bn2's apply pass is folded into conv3's staging (the implicit-GEMM kernels take in_scale / in_shift too)
            cp3 = self.gn_then_conv(S, t2, p + ".bn2", RELU, cp, p + ".conv3", t3)
        else:
            self.gn(S, t2, p + ".bn2", RELU, conv_p=cp)
            cp3 = self.conv(S, t2, p + ".conv3", t3)
        res_norm = None
        if has_ds:
            idt = self.buf("btd", Ho, Wo, planes * 4)
            cp = self.conv(S, x, p + ".downsample.0", idt, stride=stride)
            if FUSE_GN_APPLY and FUSE_GN_STATS:
                # round 3: the identity path's GroupNorm (no activation) has ONE reader, the residual input of bn3's apply
                # pass below: it is normalised there on the fly (res_scale / res_shift from otvm_gn_table) -- the four widest
                # apply passes of the encoder (256 / 512 / 1024 / 2048 channels) are gone, the tensor stays raw in memory
                sc, sh, nbs = self.gn_table_step(S, idt, p + ".downsample.1", cp)
                res_norm = (sc, sh, NONE, nbs)
            else:
                self.gn(S, idt, p + ".downsample.1", NONE, conv_p=cp)
        else:
            idt = x
        self.gn(S, t3, p + ".bn3", RELU, out=out, residual=idt, conv_p=cp3, res_norm=res_norm)

    def _predicted_tail(self, S, x, p, planes, stride, has_ds, out, t2, cp2):
        """Round 4: bn2 -> conv3 -> bn3 -> (+ identity) -> ReLU of an FBA bottleneck with bn3's statistics PREDICTED from conv3's
        input (csrc/gram.hip), so conv3's epilogue writes the block output and the apply pass is gone.  Returns False when the
        block keeps round 3's route (switched off, exact-fp32 path, a conv3 the tables were not built for)."""
        e, lib, sd = self.e, self.lib, self.e.sd
        wname = p + ".conv3"
        gp = e.GP.get(wname) if hasattr(e, "GP") else None
        if not (FUSE_GN_PREDICT and not e.gn_predict_off and gp is not None and FUSE_GN_APPLY and FUSE_GN_STATS and FUSE_GN_TABLE and FUSE_GN_APPLY_IGEMM
                and cp2 is not None and (GN_PREDICT_DS or not has_ds) and t2.P >= GN_PREDICT_MIN_PIXELS
                and planes in GN_PREDICT_PLANES):
            return False
        w = e.W[wname]
        probe = conv_params(t2, w, out, None, 1, 0, 1, RELU, 0, x, e.conv_precision, (1, 1, RELU))
        if lib.otvm_conv2d_input_norm_kind(C.byref(probe)) != 2:
            return False
        B, C4 = self.B, planes * 4
        Ho, Wo = t2.H, t2.W
        sc, sh, nbs = self.gn_table_step(S, t2, p + ".bn2", cp2)         # bn2's table, written by conv2's last workgroup
        rsc = rsh = None
        if has_ds:
            # the identity path first: its GroupNorm (no activation) stays unapplied -- conv3's epilogue multiplies the raw
            # tensor by its scale (otvm_conv_params.res_scale), the shift joins conv3's bias (otvm_gn_predict.res_shift)
            idt = self.buf("btd", Ho, Wo, C4)
            cpd = self.conv(S, x, p + ".downsample.0", idt, stride=stride)
            rsc, rsh, rnbs = self.gn_table_step(S, idt, p + ".downsample.1", cpd)
        else:
            idt = x
        nk = int(lib.otvm_gram_chunks(t2.P, planes, None))
        ent = int(lib.otvm_gram_entries(planes))
        gpart = self.raw("gram_g", B * nk * ent)                         # shared by all blocks of a size (a chain)
        spart = self.raw("gram_s", B * nk * planes)
        q = L.GramParams()
        q.x, q.P, q.C, q.ld = t2.ptr, t2.P, planes, t2.ld
        q.in_scale, q.in_shift, q.in_act = sc, sh, RELU
        q.gpart, q.spart, q.passes = gpart.data_ptr(), spart.data_ptr(), GN_PREDICT_PASSES
        q.batch, q.x_bs, q.norm_bs = B, t2.bs, nbs
        if self.diag is None:
            self.diag = self.raw("gnpred_diag", 2 * 64, torch.int32)     # [layer][kappa bits, flags] (ABI 18)
        slot = len(self._predicted)
        assert slot < 64
        q.diag = self.diag.data_ptr() + 8 * slot
        self._predicted.append((p, q, slot))
        self._keep.append(q)
        S.append((lib.otvm_gram_f16, (C.byref(q),), "gram " + p))
        tab = self.raw("gnpred_" + p, 2 * C4 * B)                        # [B][scale_eff C4 | bias_eff C4]
        cnt = self.raw("gnpredcnt_" + p, B, torch.int32)
        r = L.GnPredictParams()
        r.gpart, r.spart, r.P, r.C, r.Cout = gpart.data_ptr(), spart.data_ptr(), t2.P, planes, C4
        r.Mp, r.v = gp[0].data_ptr(), gp[1].data_ptr()
        r.counter = cnt.data_ptr()
        r.ws = self.raw("gnpred_ws", B * int(lib.otvm_gn_predict_ws_bytes()) // 8, torch.float64).data_ptr()   # (launches are serial)
        r.wscale, r.gamma, r.beta = w.w_scale.data_ptr(), sd[p + ".bn3.weight"].data_ptr(), sd[p + ".bn3.bias"].data_ptr()
        r.res_shift = 0 if rsh is None else rsh
        r.scale_eff, r.bias_eff = tab.data_ptr(), tab.data_ptr() + 4 * C4
        r.batch, r.tab_bs, r.rs_bs = B, 2 * C4, 0 if rsh is None else rnbs
        r.diag = q.diag
        self._keep.append(r)
        S.append((lib.otvm_gn_predict, (C.byref(r),), "gn_predict " + p))
        cp3 = self.conv(S, t2, wname, out, in_norm=(sc, sh, RELU, nbs), residual=idt, act=RELU)
        cp3.w_scale, cp3.bias, cp3.ws_bs = tab.data_ptr(), tab.data_ptr() + 4 * C4, 2 * C4
        if rsc is not None:
            cp3.res_scale, cp3.res_scale_bs = rsc, rnbs
        return True

    def bn_bottleneck(self, S, x, p, planes, stride, has_ds, out, tag):
        Ho, Wo = x.H // stride, x.W // stride
        W = self.e.W
        c3 = W.get(p + (".conv3cat" if has_ds else ".conv3"))
        if (FUSE_STM_BLOCK and self.e.precision == L.PREC_F16X3 and planes == 64 and stride == 1 and c3 is not None
                and c3.w_wfrag is not None and x.C == (64 if has_ds else 256)):
            # one launch, t1 / t2 never leave the CU (csrc/bottleneck_f16x3.hip)
            c1, c2 = W[p + ".conv1"], W[p + ".conv2"]
            q = L.StmBottleneckParams(x.ptr, x.H, x.W, x.C, x.ld, out.ptr, out.ld,
                                      c1.w_wfrag.data_ptr(), c2.w_wfrag.data_ptr(), c3.w_wfrag.data_ptr(),
                                      c1.w_scale.data_ptr(), c2.w_scale.data_ptr(), c3.w_scale.data_ptr(),
                                      c1.bias.data_ptr(), c2.bias.data_ptr(), c3.bias.data_ptr(), x.B, x.bs, out.bs, 0)
            self._keep.append(q)
            P = x.B * x.H * x.W
            flops = 2 * P * (x.C * 64 + 9 * 64 * 64 + 64 * 256 + (x.C * 256 if has_ds else 0))
            abytes = 4 * (P * x.C + P * 256 + x.C * 64 + 9 * 64 * 64 + 64 * 256 + (x.C * 256 if has_ds else 0))
            S.append((self.lib.otvm_stm_bottleneck_f16x3, (C.byref(q),), "conv " + p + " (fused bottleneck)", flops, abytes,
                      (x, out.ch(0, 256) if out.C > 256 else out)))
            return
        c1, c2, c3 = W[p + ".conv1"], W[p + ".conv2"], W.get(p + ".conv3")
        fuse128 = (FUSE_STM_BLOCK128 and self.e.precision == L.PREC_F16X3 and self.e.conv_precision == L.PREC_F16X3 and planes == 128
                   and stride == 1 and not has_ds and x.C == 512 and c3 is not None
                   and all(c.w_wfrag is not None and c.bias is not None for c in (c1, c2, c3))
                   and x.H * x.W * x.ld * 4 < (1 << 31))
        U = [] if fuse128 else S                                 # the three launches (the alternative of a fused block)
        t1 = self.buf(tag + "t1", x.H, x.W, planes)
        self.conv(U, x, p + ".conv1", t1, act=RELU)
        t2 = self.buf(tag + "t2", Ho, Wo, planes)
        self.conv(U, t1, p + ".conv2", t2, stride=stride, pad=1, act=RELU)
        if has_ds:
            idt = self.buf(tag + "td", Ho, Wo, planes * 4)
            self.conv(U, x, p + ".downsample.0", idt, stride=stride)
        else:
            idt = x
        self.conv(U, t2, p + ".conv3", out, residual=idt, act=RELU)
        if fuse128:
            q = L.StmBottleneckParams(x.ptr, x.H, x.W, x.C, x.ld, out.ptr, out.ld,
                                      c1.w_wfrag.data_ptr(), c2.w_wfrag.data_ptr(), c3.w_wfrag.data_ptr(),
                                      c1.w_scale.data_ptr(), c2.w_scale.data_ptr(), c3.w_scale.data_ptr(),
                                      c1.bias.data_ptr(), c2.bias.data_ptr(), c3.bias.data_ptr(), x.B, x.bs, out.bs, 0)
            self._keep.append(q)
            P = x.B * x.H * x.W
            flops = 2 * P * (512 * 128 + 9 * 128 * 128 + 128 * 512)
            abytes = 4 * (P * 512 * 2 + 512 * 128 * 2 + 9 * 128 * 128)
            fused = (self.lib.otvm_stm_bottleneck_f16x3, (C.byref(q),), "conv " + p + " (fused bottleneck)", flops, abytes,
                     (x, out.ch(0, 512) if out.C > 512 else out))
            # resolved by _resolve_stm128 (timed at plan time): either `fused` or the launches of `U` take this place
            S.append(("stm128", fused, U, q, p))
            self._stm128.append(S)

    def stm_trunk(self, S, stem_out, e, tag):
        """maxpool + res2/res3/res4 (BN folded) of an STM encoder.  Returns r4, r3, r2."""
        H4, W4 = self.Hp // 4, self.Wp // 4
        x = self.buf(tag + "pool", H4, W4, 64)
        self.maxpool(S, stem_out, x)
        outs = {}
        for lname, n, planes, s0 in (("res2", 3, 64, 1), ("res3", 4, 128, 2), ("res4", 6, 256, 2)):
            for b in range(n):
                st = s0 if b == 0 else 1
                o = self.buf(tag + lname + ("a" if b % 2 == 0 else "b"), x.H // st, x.W // st, planes * 4)
                self.bn_bottleneck(S, x, e + "%s.%d" % (lname, b), planes, st, b == 0, o, tag)
                x = o
            outs[lname] = x
        return outs["res4"], outs["res3"], outs["res2"]

    def resblock(self, S, x, p, out, tag):
        """STM.py:9-30 (no downsample case): out = x + conv2(relu(conv1(relu(x))))."""
        r = self.buf(tag + "r", x.H, x.W, 256)
        self.conv(S, x, p + ".conv1", r, pad=1, in_relu=1)
        self.conv(S, r, p + ".conv2", out, pad=1, in_relu=1, residual=x)

    def _build(self):
        e, lib = self.e, self.lib
        Hp, Wp, P = self.Hp, self.Wp, self.P
        H2, W2, H4, W4, H8, W8, H16, W16 = Hp // 2, Wp // 2, Hp // 4, Wp // 4, Hp // 8, Wp // 8, Hp // 16, Wp // 16
        self.hw = H16 * W16

        # split-K workspaces, one per chain that may run concurrently with the others: [0] decoder + alpha network (launch
        # stream), [1] memorize (launch stream, but tuned / timed separately), [2] query encoder (side stream)
        self.SPLITK_WS = [self.raw("splitk_ws%d" % i, (16 << 20) * self.B) for i in range(4)]    # [3]: decoder skip branches (side stream 2)
        self._ws = self.SPLITK_WS[2]
        # ---------------- frame-level buffers
        self.X11 = self.buf("X11", Hp, Wp, 12)          # 0-2 normalised RGB, 3-8 distance encoding, 9-10 soft, 11 zero
        self.SQ = self.buf("SQ", Hp, Wp, 4)             # Encoder_Q input (normalised RGB)
        # Encoder_M input: hid16 | rgb | p_un p_fg alpha | pad.  Two copies (frame parity): frame t's memorize is
        # deferred to the start of frame t+1 and overlaps Encoder_Q(t+1) on a second stream, so SM(t) must survive
        # while frame t+1 writes its own.
        self.SMs = [self.buf("SM0", Hp, Wp, 24), self.buf("SM1", Hp, Wp, 24)]
        self.D80 = self.buf("D80", Hp, Wp, 80)          # conv_up3 out 0-63 | rgb_n 64-66 | rgb 67-69 | tri2 70-71 | alpha 72
        # planar per-image tensors of the glue kernels (one per image of the batch; the un-suffixed names = image 0)
        self.PROBS_B = self.raws("probs", 3 * P)        # planar trimap probabilities fed to the encoding
        self.CLS_B = self.raws("cls", P, torch.uint8)
        self.ALPHA_P_B = self.raws("alpha_p", P)
        self.TRI_P_B = self.raws("tri_p", 3 * P)
        self.enc_ws_B = self.raws("enc_ws", _rup(int(lib.otvm_trimap_encode_ws_bytes(Hp, Wp)), 256), torch.uint8)
        self.PROBS, self.CLS, self.ALPHA_P, self.TRI_P = self.PROBS_B[0], self.CLS_B[0], self.ALPHA_P_B[0], self.TRI_P_B[0]

        # ---------------- STM segment (STM.py:239-257)
        S = []
        sd = e.sd
        q = "trimap.model.Encoder_Q."
        stem = self.buf("q_stem", H2, W2, 64)
        self.conv(S, self.SQ, q + "conv1", stem, stride=2, pad=3, act=RELU)
        r4, r3, r2 = self.stm_trunk(S, stem, q, "q_")
        self.QK = self.buf("QK", H16, W16, 128)
        self.M4 = self.buf("M4", H16, W16, 1024)
        self.conv(S, r4, "trimap.model.KV_Q_r4.Key", self.QK, pad=1)
        self.conv(S, r4, "trimap.model.KV_Q_r4.Value", self.M4.ch(512, 512), pad=1)
        self.steps["segment_a"] = S
        self._ws = self.SPLITK_WS[0]
        S = []
        d = "trimap.model.Decoder."
        # The skip branches of the two Refine blocks (STM.py:110-113: ResFS(convFS(r3 / r2))) read only the query
        # encoder's features, not the memory readout: they form their own launch list ("segment_skip", 1.7 of the
        # decoder's 3.1 ms at 1080p), which the engine runs on a second side stream next to Encoder_M of the previous
        # frame -- dense kernels filling the CUs that chain of small launches leaves idle.
        SK = []
        self._ws = self.SPLITK_WS[3]
        skips = {}
        for rf, feat, (h, w) in (("RF3", r3, (H8, W8)), ("RF2", r2, (H4, W4))):
            s0 = self.buf("d_s0", h, w, 256)
            self.conv(SK, feat, d + rf + ".convFS", s0, pad=1)
            s1 = self.buf("d_s1", h, w, 256)
            self.resblock(SK, s0, d + rf + ".ResFS", s1, "fs%d" % h)
            skips[rf] = s1
        self.steps["segment_skip"] = SK
        self._ws = self.SPLITK_WS[0]
        m = self.buf("d_m4a", H16, W16, 256)
        self.conv(S, self.M4, d + "convFM", m, pad=1)
        m4 = self.buf("d_m4b", H16, W16, 256)
        self.resblock(S, m, d + "ResMM", m4, "d16")
        pm = m4
        for rf, (h, w) in (("RF3", (H8, W8)), ("RF2", (H4, W4))):
            mm = self.buf("d_mm", h, w, 256)
            self.upsample(S, pm, mm, add=skips[rf])                 # m = s + up2(pm)  (STM.py:115)
            mo = self.buf("d_mo", h, w, 256)
            self.resblock(S, mm, d + rf + ".ResMM", mo, "d%d" % h)
            pm = mo
        self.L4 = self.buf("L4", H4, W4, 4)
        self.conv(S, pm, d + "pred", self.L4, pad=1, in_relu=1)
        for b in range(self.B):
            S.append((lib.otvm_upsample4_softmax3, (self.L4.img(b).ptr, H4, W4, self.L4.ld, self.PROBS_B[b].data_ptr()), "up4softmax"))
        self.steps["segment_b"] = S

        # ---------------- FBA encoder (FBA/models.py:251-269)
        S = []
        en = "NET.encoder."
        self.U3 = self.buf("U3", H2, W2, 320)            # [up(conv_up2) 256 | c1 64]
        self.U2 = self.buf("U2", H4, W4, 512)            # [up(conv_up1) 256 | l1 256]
        # [l4 2048 | ppm 4x256]; with the PPM algebra (round 3) the upsampled PPM maps do not exist: layer 4 alone
        ppm_alg = FUSE_PPM_HEAD and self.e.W_ppm is not None
        self.PPMCAT = self.buf("PPMCAT", H8, W8, 2048 if ppm_alg else 3072)
        c1raw = self.buf("c1raw", H2, W2, 64)
        cp = self.conv(S, self.X11, en + "conv1", c1raw, stride=2, pad=3)
        c1 = self.U3.ch(256, 64)
        self.gn(S, c1raw, en + "bn1", RELU, out=c1, conv_p=cp)
        x = self.buf("e_pool", H4, W4, 64)
        self.maxpool(S, c1, x)
        cfg = {"layer1": (64, 3, 1, 1, 1), "layer2": (128, 4, 2, 1, 1), "layer3": (256, 6, 1, 1, 2),
               "layer4": (512, 3, 1, 2, 4)}
        for lname in ("layer1", "layer2", "layer3", "layer4"):
            planes, n, s0, d0, dn = cfg[lname]
            for b in range(n):
                st = s0 if b == 0 else 1
                last = b == n - 1
                if last and lname == "layer1":
                    o = self.U2.ch(256, 256)
                elif last and lname == "layer4":
                    o = self.PPMCAT.ch(0, 2048)
                else:
                    o = self.buf("e_" + lname + ("a" if b % 2 == 0 else "b"), x.H // st, x.W // st, planes * 4)
                self.gn_bottleneck(S, x, en + "%s.%d" % (lname, b), planes, st, d0 if b == 0 else dn, b == 0, o)
                x = o
        # ---------------- FBA decoder (FBA/models.py:351-392)
        de = "NET.decoder."
        conv5 = self.PPMCAT.ch(0, 2048)
        self.POOL_B = self.raws("ppm_pool", 50 * 2048)
        self.POOL_WS_B = self.raws("ppm_ws", int(lib.otvm_ppm_pool_ws_bytes(H8, 2048)) // 4)
        for b in range(self.B):                          # (three small launches per image: not batched)
            S.append((lib.otvm_ppm_pool, (conv5.img(b).ptr, H8, W8, 2048, conv5.ld, self.POOL_B[b].data_ptr(),
                                          self.POOL_WS_B[b].data_ptr()), "ppm_pool"))
        if FUSE_PPM_HEAD and self.e.W[de + "ppm.0.1"].I_pad == 2048 and self.e.W[de + "ppm.0.1"].O == 256:
            # conv + bias + GroupNorm + LeakyReLU of the four pooled maps in ONE launch (16 launches as library calls)
            sd = self.e.sd
            ys = [self.buf("ppm_y%d" % i, s_, s_, 256) for i, s_ in enumerate((1, 2, 3, 6))]
            for b in range(self.B):
                hp = L.PpmHeadParams()
                hp.pooled, hp.C, hp.Cout, hp.act = self.POOL_B[b].data_ptr(), 2048, 256, LEAKY
                for i in range(4):
                    w = self.e.W[de + "ppm.%d.1" % i]
                    hp.K_pad, hp.out_ld = w.K_pad, ys[i].ld
                    hp.w[i], hp.bias[i] = w.w.data_ptr(), 0 if w.bias is None else w.bias.data_ptr()
                    hp.gamma[i], hp.beta[i] = sd[de + "ppm.%d.2.weight" % i].data_ptr(), sd[de + "ppm.%d.2.bias" % i].data_ptr()
                    hp.out[i] = ys[i].img(b).ptr
                self._keep.append(hp)
                S.append((lib.otvm_ppm_head, (C.byref(hp),), "ppm_head"))
            self._ppm_algebra = ppm_alg
            if not self._ppm_algebra:
                for i, y in enumerate(ys):
                    self.upsample(S, y, self.PPMCAT.ch(2048 + 256 * i, 256))
        else:
            self._ppm_algebra = False
            if self.B > 1:
                raise NotImplementedError("otvm_amd: batched sequences need the fused PPM head (OTVM_PPM_HEAD=1)")
            base = 0
            for i, s_ in enumerate((1, 2, 3, 6)):
                pin = Act(self.POOL_B[0], s_, s_, 2048, 2048, base * 2048)
                y = self.buf("ppm_y%d" % i, s_, s_, 256)
                cp = self.conv(S, pin, de + "ppm.%d.1" % i, y)
                self.gn_then_upsample(S, y, de + "ppm.%d.2" % i, LEAKY, cp, self.PPMCAT.ch(2048 + 256 * i, 256))
                base += s_ * s_
        u1 = self.buf("u1a", H8, W8, 256)
        if self._ppm_algebra:
            # the PPM maps are never upsampled: conv_up1.0 over layer 4 alone, then the PPM channels' share of the same
            # convolution from the 50 pooled pixels (resample.hip: otvm_ppm_conv_z / _add), then the GroupNorm statistics
            self.PPM_Z = self.raws("ppm_z", 9 * 50 * 256)
            for b in range(self.B):
                yp = (C.c_void_p * 4)(*[y.img(b).ptr for y in ys])
                self._keep.append(yp)
                S.append((lib.otvm_ppm_conv_z, (yp, ys[0].ld, self.e.W_ppm.data_ptr(), self.PPM_Z[b].data_ptr()), "ppm_conv_z"))
            self.conv(S, self.PPMCAT.ch(0, 2048), de + "conv_up1.0.main", u1, pad=1)
            for b in range(self.B):
                S.append(("ppm_add", (self.PPM_Z[b].data_ptr(), H8, W8, u1.img(b).ptr, u1.ld), b, "ppm_conv_add"))
            # the gather writes the layer's final values: it also accumulates their GroupNorm sums (bound in _bind_stats)
            import types
            self._ppm_stats = types.SimpleNamespace(gn_stats=0, gn_bs=0)
            self.gn(S, u1, de + "conv_up1.1", LEAKY, conv_p=self._ppm_stats)
        else:
            cp = self.conv(S, self.PPMCAT, de + "conv_up1.0", u1, pad=1)
            self.gn(S, u1, de + "conv_up1.1", LEAKY, conv_p=cp)
        u1b = self.buf("u1b", H8, W8, 256)
        cp = self.conv(S, u1, de + "conv_up1.3", u1b, pad=1)
        self.gn_then_upsample(S, u1b, de + "conv_up1.4", LEAKY, cp, self.U2.ch(0, 256))
        u2 = self.buf("u2", H4, W4, 256)
        cp = self.conv(S, self.U2, de + "conv_up2.0", u2, pad=1)
        self.gn_then_upsample(S, u2, de + "conv_up2.1", LEAKY, cp, self.U3.ch(0, 256))
        u3 = self.buf("u3", H2, W2, 64)
        cp = self.conv(S, self.U3, de + "conv_up3.0", u3, pad=1)
        self.gn_then_upsample(S, u3, de + "conv_up3.1", LEAKY, cp, self.D80.ch(0, 64))
        h32 = self.buf("h32", Hp, Wp, 32)
        self.conv(S, self.D80, de + "conv_up4.0", h32, pad=1, act=LEAKY)      # ch 72.. carry zero weights
        img = self.D80.ch(67, 3)
        fuse_head = FUSE_HEAD and e.precision == L.PREC_F16X3 and e.W[de + "conv_up4.2"].w_frag is not None
        if fuse_head:
            hid_d = self.HID_D = self.buf("hid_d", Hp, Wp, 16) if e.keep_hid_d else None
            self.conv_head(S, h32, de + "conv_up4.2", hid_d, sd[de + "conv_up4.4.weight"], sd[de + "conv_up4.4.bias"], 7, img,
                           self.D80.ch(72, 1).ptr, self.D80.ld, self.D80.bs)
        else:
            hid_d = self.HID_D = self.buf("hid_d", Hp, Wp, 16)
            self.conv(S, h32, de + "conv_up4.2", hid_d, pad=1, act=LEAKY)
            for b in range(self.B):
                S.append((lib.otvm_fba_head,
                          (hid_d.img(b).ptr, hid_d.ld, sd[de + "conv_up4.4.weight"].data_ptr(), sd[de + "conv_up4.4.bias"].data_ptr(), 7,
                           img.img(b).ptr, img.ld, P, self.D80.ch(72, 1).img(b).ptr, self.D80.ld, 0, 0, 0), "fba_head7"))
        # ---------------- refinement (FBA/models.py:417-435)
        rf = "NET.refine."
        r0 = self.buf("r0", Hp, Wp, 64)
        cp = self.conv(S, self.D80, rf + "conv1.0", r0, pad=1)
        # conv1's GroupNorm + LeakyReLU is read twice, by layer1.conv1 (a patch conv: normalises while staging) and as the
        # residual of layer1.bn2's apply pass (normalises on the fly): the normalised 535 MB tensor is never written
        x, x_norm = r0, None
        w1 = e.W[rf + "layer1.conv1"]
        t1 = self.buf("rt1", Hp, Wp, 64)
        probe = conv_params(r0, w1, t1, w1.bias, 1, 1, 1, NONE, 0, None, e.conv_precision, (1, 1, LEAKY))
        if FUSE_GN_APPLY and FUSE_GN_STATS and lib.otvm_conv2d_accepts_input_norm(C.byref(probe)):
            sc, sh, nbs = self.gn_table_step(S, r0, rf + "conv1.1", cp)
            x_norm = (sc, sh, LEAKY, nbs)
        else:
            self.gn(S, r0, rf + "conv1.1", LEAKY, conv_p=cp)
        tail = None
        for l in ("layer1", "layer2"):
            cp = self.conv(S, x, rf + l + ".conv1", t1, pad=1, in_norm=x_norm)
            t2 = self.buf("rt2", Hp, Wp, 64)
            cp = self.gn_then_conv(S, t1, rf + l + ".bn1", RELU, cp, rf + l + ".conv2", t2, pad=1)
            if l == "layer2" and FUSE_REFINE_TAIL and FUSE_GN_APPLY and FUSE_GN_STATS and x_norm is None:
                # round 4: layer2's bn2 -> (+ identity) -> ReLU has ONE reader, pred.0: folded into its staging when the
                # library takes it (otvm_conv_params.in_res) -- the block output is never written
                w0 = e.W[rf + "pred.0"]
                probe = conv_params(t2, w0, h32, w0.bias, 1, 1, 1, LEAKY, 0, None, e.conv_precision, (1, 1, RELU))
                if lib.otvm_conv2d_accepts_input_residual(C.byref(probe)):
                    sc, sh, nbs = self.gn_table_step(S, t2, rf + l + ".bn2", cp)
                    tail = (t2, (sc, sh, RELU, nbs), x)
                    break
            o = self.buf("r_" + l, Hp, Wp, 64)
            self.gn(S, t2, rf + l + ".bn2", RELU, out=o, residual=x, conv_p=cp, res_norm=x_norm)
            x, x_norm = o, None
        if tail is not None:
            self.conv(S, tail[0], rf + "pred.0", h32, pad=1, act=LEAKY, in_norm=tail[1], in_res=tail[2])
        else:
            self.conv(S, x, rf + "pred.0", h32, pad=1, act=LEAKY)
        self.steps["fba"] = S
        for par in (0, 1):
            S = []
            SM = self.SMs[par]
            hid = SM.ch(0, 16)
            if fuse_head:
                self.conv_head(S, h32, rf + "pred.2", hid, sd[rf + "pred.4.weight"], sd[rf + "pred.4.bias"], 10, img,
                               self.ALPHA_P_B[0].data_ptr(), 1, P, tri=self.TRI_P_B[0].data_ptr(), sm=SM.ch(16, 8))
                self.steps["fba_tail%d" % par] = S
                continue
            self.conv(S, h32, rf + "pred.2", hid, pad=1, act=LEAKY)
            for b in range(self.B):
                S.append((lib.otvm_fba_head,
                          (hid.img(b).ptr, hid.ld, sd[rf + "pred.4.weight"].data_ptr(), sd[rf + "pred.4.bias"].data_ptr(), 10,
                           img.img(b).ptr, img.ld, P, self.ALPHA_P_B[b].data_ptr(), 1, self.TRI_P_B[b].data_ptr(),
                           SM.ch(16, 8).img(b).ptr, SM.ld), "fba_head10"))
            self.steps["fba_tail%d" % par] = S

        # ---------------- STM memorize (STM.py:201-228); key/value convs are bound to a slot at run time
        self._ws = self.SPLITK_WS[1]
        m_ = "trimap.model.Encoder_M."
        stem = self.buf("m_stem", H2, W2, 64)
        for par in (0, 1):
            S = []
            self.conv(S, self.SMs[par], m_ + "stem", stem, stride=2, pad=3, act=RELU)
            self.steps["mem_stem%d" % par] = S
        S = []
        self.r4m, _, _ = self.stm_trunk(S, stem, m_, "m_")
        self.steps["mem_trunk"] = S
        self.mem_ws = None

    # ------------------------------------------------------------------ run
    def run(self, key, stream, tstream=None):
        """Launch the steps of list ``key`` on ``stream`` (raw hipStream_t).  The lists are static (fixed buffers and
        parameters), so from their second use on they are replayed as ONE hipGraph each (captured through
        torch.cuda.graph; ``tstream`` = the torch stream object when it is not the current one): ~300 kernel launches per
        frame become a handful of graph launches (host time per frame 6.8 -> 0.9 ms at 480p; no throughput change, every
        measured configuration is GPU-bound).  Opt-in: OTVM_GRAPHS=1 or engine.use_graphs = True."""
        prof = self.e.prof
        nc = self._n_conv.get(key)
        if nc is None:
            nc = self._n_conv[key] = sum(1 for st in self.steps[key] if not isinstance(st[0], str) and st[2].startswith("conv "))
        self.e.conv_calls += nc
        if prof is None and self.e.check_level >= 3:
            # first run of a new checkpoint: scan the input and the output of every convolution
            for st in self.steps[key]:
                if isinstance(st[0], str):
                    continue                              # (fork / join markers: one serial stream here)
                if st[2].startswith("conv "):
                    self.e.guard(st[5][0], "input of " + st[2], stream)
                rc = st[0](*st[1], stream)
                if rc != 0:
                    L.check(rc, st[2])
                if st[2].startswith("conv "):
                    self.e.scan_now(st[5][1], "output of " + st[2], stream)
            return
        if prof is None:
            if graphs_wanted(self.e.use_graphs, self.P):
                g = self.graphs.get(key)
                if g is None and self._graph_warm.get(key):
                    g = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(g, capture_error_mode="thread_local"):   # other threads (the IO pipeline) keep issuing copies
                        self._launch(self.steps[key], torch.cuda.current_stream(self.dev))
                    self.graphs[key] = g
                if g is not None:
                    if tstream is None:
                        g.replay()
                    else:
                        with torch.cuda.stream(tstream):
                            g.replay()
                    return
                self._graph_warm[key] = True              # first use: direct launches (module loading, warm-up)
            self._launch(self.steps[key], tstream if tstream is not None else torch.cuda.current_stream(self.dev), stream)
            return
        # instrumented pass (bench.py roofline leg): HIP events around every conv launch, on this stream
        for st in self.steps[key]:
            if isinstance(st[0], str):
                continue
            if st[2].startswith("conv "):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                rc = st[0](*st[1], stream)
                e1.record()
                prof.append((st[2], st[3], e0, e1, st[4]))
            else:
                rc = st[0](*st[1], stream)
            if rc != 0:
                L.check(rc, st[2])

    def _launch(self, steps, tmain, handle=None):
        """Issue ``steps`` on the torch stream ``tmain`` (raw handle ``handle``).  Every step must be a bound library call by now:
        a string marker left in a list (a step _bind_stats did not rebind) is an error, not something to skip."""
        handle = tmain.cuda_stream if handle is None else handle
        for st in steps:
            f = st[0]
            if isinstance(f, str):
                raise RuntimeError("otvm_amd: unbound step %r in a launch list" % (f,))
            rc = f(*st[1], handle)
            if rc != 0:
                L.check(rc, st[2])

    def new_slot(self):
        e = self.e
        # a slot carries launch parameters bound to THIS plan's buffers (kv_steps read self.r4m): it may only be recycled
        # by the plan that made it -- two input sizes with the same padded size have different plans and buffers
        for i, s in enumerate(e.free_slots):
            if s["plan"] is self:
                return e.free_slots.pop(i)
        H16, W16, B = self.Hp // 16, self.Wp // 16, self.B
        slot = dict(hw=self.hw, plan=self, frame=-1,
                    k=Act(torch.zeros(B * self.hw * 128, dtype=torch.float32, device=self.dev), H16, W16, 128, B=B, bs=self.hw * 128),
                    v=Act(torch.zeros(B * self.hw * 512, dtype=torch.float32, device=self.dev), H16, W16, 512, B=B, bs=self.hw * 512))
        if e.precision == L.PREC_F16X3:        # packed (split fp16, MFMA fragment order) copy read by the f16x3 kernel, per image
            nb = int(self.lib.otvm_bank_slot_bytes_f16x3(self.hw))
            slot["packed_b"] = [torch.zeros(nb, dtype=torch.uint8, device=self.dev) for _ in range(B)]
            slot["packed"] = slot["packed_b"][0]
        return slot

    # ---- memory read (STM.py:140-163).  Every image of a batch has its own bank (per-sequence slots) and its own
    # workspace: the read is launched per image (three launches of ~320 per frame; the rest of the frame is batched)
    def _mem_ws_for(self, b, need):
        if self.mem_ws is None:
            self.mem_ws = [None] * self.B
        w = self.mem_ws[b]
        if w is None or w.numel() < need:
            torch.cuda.synchronize(self.dev)                  # (rare: the bank grew; nothing in flight may use the old one)
            w = self.mem_ws[b] = torch.empty(max(need, 2 * (0 if w is None else w.numel())), dtype=torch.uint8, device=self.dev)
        return w

    def memory_read(self, bank, stream):
        T = len(bank)
        need = int(self.lib.otvm_memory_read_ws_bytes(self.hw, T))
        for b in range(self.B):
            ws = self._mem_ws_for(b, need)
            out, qk = self.M4.ch(0, 512).img(b), self.QK.img(b)
            if self.e.precision == L.PREC_F16X3:
                slots = (C.c_void_p * T)(*[s["packed_b"][b].data_ptr() for s in bank])
                L.check(self.lib.otvm_memory_read_f16x3(qk.ptr, qk.ld, slots, T, self.hw, out.ptr, out.ld, ws.data_ptr(), stream),
                        "memory_read_f16x3")
                continue
            keys = (C.c_void_p * T)(*[s["k"].img(b).ptr for s in bank])
            vals = (C.c_void_p * T)(*[s["v"].img(b).ptr for s in bank])
            L.check(self.lib.otvm_memory_read(qk.ptr, qk.ld, keys, vals, T, self.hw, out.ptr, out.ld, ws.data_ptr(), stream),
                    "memory_read")

    def memory_read_begin(self, old, fresh, stream):
        """f16x3 memory read, first step: partials over the slots ``old`` (already resident) on ``stream``; ``fresh`` (the
        slot being memorised right now, 0 or 1 entries) follows in memory_read_fresh."""
        lib = self.lib
        np_cap = int(lib.otvm_memory_read_f16x3_partial_count(len(old), self.hw))
        if fresh:
            np_cap += int(lib.otvm_memory_read_f16x3_partial_count(len(fresh), self.hw))
        need = np_cap * self.hw * (512 + 2) * 4
        wss, done = [], 0
        for b in range(self.B):
            ws = self._mem_ws_for(b, need)
            qk = self.QK.img(b)
            arr = (C.c_void_p * len(old))(*[s["packed_b"][b].data_ptr() for s in old])
            end = C.c_int(0)
            L.check(lib.otvm_memory_read_f16x3_partial(qk.ptr, qk.ld, arr, len(old), self.hw, ws.data_ptr(), np_cap, 0, C.byref(end),
                                                       stream), "memory_read_f16x3_partial")
            wss.append(ws)
            done = end.value
        return dict(old=old, fresh=fresh, np_cap=np_cap, done=done, ws=wss)

    def memory_read_fresh(self, split, stream):
        lib, n = self.lib, split["done"]
        if split["fresh"]:
            fr = split["fresh"]
            for b in range(self.B):
                qk = self.QK.img(b)
                arr = (C.c_void_p * len(fr))(*[s["packed_b"][b].data_ptr() for s in fr])
                end = C.c_int(0)
                L.check(lib.otvm_memory_read_f16x3_partial(qk.ptr, qk.ld, arr, len(fr), self.hw, split["ws"][b].data_ptr(),
                                                           split["np_cap"], n, C.byref(end), stream), "memory_read_f16x3_partial")
            n = end.value
        split["done"] = n

    def memory_read_merge(self, split, stream):
        for b in range(self.B):
            out = self.M4.ch(0, 512).img(b)
            L.check(self.lib.otvm_memory_read_f16x3_combine(split["ws"][b].data_ptr(), split["np_cap"], split["done"], self.hw,
                                                            out.ptr, out.ld, stream), "memory_read_f16x3_combine")

    def kv_into_slot(self, slot, stream):
        if "kv_steps" not in slot:
            S = []
            n0 = len(self._convs)
            self.conv(S, self.r4m, "trimap.model.KV_M_r4.Key", slot["k"], pad=1)
            self.conv(S, self.r4m, "trimap.model.KV_M_r4.Value", slot["v"], pad=1)
            self.tune_convs(self._convs[n0:])                  # shapes timed at plan time (autotune): cache hits
            slot["kv_steps"] = S
        prof = self.e.prof
        self.e.conv_calls += len(slot["kv_steps"])
        for st in slot["kv_steps"]:
            if prof is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
            L.check(st[0](*st[1], stream), st[2])
            if prof is not None:
                e1.record()
                prof.append((st[2], st[3], e0, e1, st[4]))
            if prof is None and self.e.check_level >= 3:
                self.e.scan_now(st[5][1], "output of " + st[2], stream)
        self.e.guard(slot["k"], "memorised key", stream, slot["frame"])
        self.e.guard(slot["v"], "memorised value", stream, slot["frame"])
        if "packed_b" in slot:
            for b in range(self.B):
                L.check(self.lib.otvm_bank_pack_f16x3(slot["k"].img(b).ptr, slot["v"].img(b).ptr, self.hw,
                                                      slot["packed_b"][b].data_ptr(), stream), "bank_pack")

    def encode(self, stream, cls_override=None):
        """8-channel trimap encoding of PROBS into X11[3:11] / D80[70:72] (alpha/model.py:40-53), image by image."""
        for b in range(self.B):
            co = None if cls_override is None else (cls_override[b] if isinstance(cls_override, (list, tuple)) else cls_override)
            x11, d80 = self.X11.img(b), self.D80.img(b)
            L.check(self.lib.otvm_trimap_encode(self.PROBS_B[b].data_ptr(), self.Hp, self.Wp, 0 if co is None else co.data_ptr(),
                                                self.CLS_B[b].data_ptr(), x11.ptr, x11.ld, d80.ptr, d80.ld,
                                                self.enc_ws_B[b].data_ptr(), stream), "trimap_encode")

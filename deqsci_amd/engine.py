"""DEQ-SCI reconstruction engine: the whole GAP/prox-grad fixed-point loop on one MI355X.

What the reference does per measurement (solvers/new_equilibrium_utils_yaping.py:153-189,249-281
driving solvers/equilibrium_solvers_yaping.py:397-425) as ~90 ATen launches, two permute copies
and three host syncs per iteration, this engine does as

    [K7+K3 mix_gap] -> denoiser (torch.nn on PyTorch-ROCm) -> [K4 residual_store] -> [K5+K6 solve]

on a batch of independent measurements, with

  * the loop state kept PLANAR (bsz,B,H,W) - the denoiser's (bsz*B,1,H,W) layout - so both
    per-iteration transposes of the reference disappear; only Phi / x0 in and the reconstruction out
    cross the (bsz,H,W,B) API layout, through the LDS-staged transpose kernels;
  * the history stored as F and G = F - X, Gram matrix updated one row per iteration;
  * the FFDNet sigma schedule (60/255 * 0.971^call, fp32 repeated multiply, ibid. :408-413) read
    from a device table by call index - no `self.y != y.mean()` host sync per call;
  * the relative residual polled one iteration late from pinned memory, so the host never drains
    the GPU queue (the iterate ping-pongs between two buffers, which makes the one speculative
    extra iteration harmless when the tolerance test fires);
  * for small batches, where a conv layer is a few tens of microseconds and the ~3400 launches of a reconstruction are
    what the GPU waits for, the WHOLE reconstruction (x0, every f-call, the output transpose) replayed as ONE hipGraph
    captured on the second call of a shape: same kernels, same arguments, bit-identical output.  The graph always runs
    max_iter iterations; the residual table is read afterwards and, should the tolerance test have fired earlier (it never
    does on the reference's data), the call is redone on the eager path, which stops where the reference stops.

Semantics are the reference's (same slots k % m, same bordered system, residual over the whole
batch as at :184, returned iterate = f(X_last)).  Deliberate deviations, all result-neutral for the
reference's own usage: sigma restarts at every reconstruct() call (the reference restarts when
y.mean() changes); the dead second f-call of DEQFixedPoint.forward (:271-272, only feeds the
backward hook) is skipped unless `extra_call=True`.
"""
import math
import numpy as np

import torch
import torch.nn.functional as F

from . import _hip
from ._hip import LAYOUT_BHW, LAYOUT_HWB

SIGMA0 = 60 / 255
SIGMA_DECAY = 0.971


def sigma_schedule(n):
    """sigma used by f-call c (0-based): fp32(60/255) multiplied c times by fp32(0.971) in fp32."""
    out = np.empty(n, dtype=np.float32)
    s = np.float32(SIGMA0)
    d = np.float32(SIGMA_DECAY)
    for i in range(n):
        out[i] = s
        s = np.float32(s * d)
    return out


def _exhaust(gen):
    """Run a generator to its end and return its return value (the plain-call form of the *_steps generators below)."""
    while True:
        try:
            next(gen)
        except StopIteration as done:
            return done.value


def _fold_bn(conv_w, bn):
    scale = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    return (conv_w * scale.view(-1, 1, 1, 1)).contiguous(), (bn.bias - bn.running_mean * scale).contiguous()


class _Denoiser:
    """Adapter from a reference-style denoiser plugin (nn.Module with .tag) to
    `run(z1_planar, call) -> (tensor (bsz,B,H,W), is_noise)`; dispatch as at
    solvers/equilibrium_solvers_yaping.py:402-425."""

    def __init__(self, net, fold_bn=True, channels_last=None, fused_epilogue=True, fused_edges=True, winograd=True, conv64="fast",
                 act_range="data", blk32=True, stack=True, stack_kernel="w16"):
        from .networks import FFDNet
        self.net = net
        # stack_kernel: which kernel a stack launch of FFDNet's run runs - "w16" (default): the split-fp16 arithmetic under Winograd F(2,3) x
        # direct (csrc/conv_w16.hip: a third fewer matrix-core products, p32 activations between FFDNet's first and last layer); "s16": the
        # split-fp16 direct convolution (csrc/conv_s16.hip).  The measuring f-call keeps the direct kernel; SimpleCNN's two middle layers (a run
        # shorter than STACK_MIN_LAYERS) go out as single launches of the chosen kernel.
        if stack_kernel not in ("w16", "s16"):
            raise ValueError(f"stack_kernel={stack_kernel!r}: expected 'w16' or 's16'")
        self.stack_kernel = stack_kernel
        self.gate = None                                            # (_StackGate of a grouped reconstruction: one stack launch on the device at a time)
        # set by the engine while it keeps off the stack launches after a time-out: FFDNet's run as ONE LAUNCH PER LAYER of the same Winograd
        # kernel - bit-identical to the stack launch (test_wino16_stack_is_bit_identical_to_single_launches), so that the calls behind a
        # time-out return the bits of ordinary calls.  (A caller's own stack=False keeps its documented meaning: the direct kernel per layer.)
        self.per_layer_w16 = False
        self.last_path = None                                       # what the last f-call's run of 64->64 layers went out as (last_info["denoiser_path"])
        self._wstacks = {}                                          # first layer of a run -> _hip.Wino16Stack
        # stack: a run of split-fp16 64->64 layers (FFDNet's 13, SimpleCNN's 2) as ONE launch per slice of the batch
        # (_hip.conv3x3_c64_split16_stack: tiles synchronised by per-tile progress words instead of kernel boundaries, slices that keep
        # their activations in the Infinity Cache); bit-identical to the per-layer launches, which the measuring f-call keeps
        self.stack = bool(stack)
        self.stack_per_launch = None                                # images per stack launch (None: _hip.split16_stack_per_launch; an A/B of the tools)
        self.slice_edges = True                                     # FFDNet: first layer -> run -> last layer slice by slice (False: an A/B of the tools)
        self._stacks = {}                                           # first layer of a run -> _hip.Split16Stack
        self.stack_launches = 0
        self.fused_edges = fused_edges
        self.winograd = winograd
        self.conv64 = conv64                                        # _hip.conv64_kernel_for policy: "fast" | "fast32" | "f22" | "f44" | "s16"
        self.f22_calls = None                                       # f-calls [0, f22_calls) run F(2x2,3x3) whatever the policy (DEQSCIEngine)
        self._native_out = False
        self._policy = conv64
        self.blk32 = bool(blk32)                                    # blk32 activations between F(4x4,3x3) layers (False: an A/B of the tools)
        # The scales of the split-fp16 layers' activations: "data" = every activation's power-of-two scale follows max |activation| as
        # measured ON THE DEVICE during the first f-call of a reconstruction (run(..., calibrate=True): each layer is run once to measure,
        # once to write - fp32 is scale-free, fp16 is not; see _hip.act_exp); "fixed" = 2^8 throughout (activations of a few units).
        if act_range not in ("data", "fixed"):
            raise ValueError(f"act_range={act_range!r}: expected 'data' or 'fixed'")
        self.act_range = act_range
        self.ranges = None                                          # (len(layers) + 1, images) fp32 on the device: [i, j] = max |input of layer i, image j|
        self._calibrating = False
        self._stale = True                                          # the ranges have not been measured on the current input yet
        # channels_last is what the HIP Winograd / edge kernels consume.  Without them (winograd=False) it is a
        # MIOpen trade-off measured on MI355X (profiles/r01_denoiser_variants.jsonl): 8 % faster for FFDNet's
        # 128x128x64 layers, 11 % slower for SimpleCNN's 256x256x64 ones.
        self.channels_last = (winograd or isinstance(net, FFDNet)) if channels_last is None else bool(channels_last)
        self.fused_epilogue = fused_epilogue
        self.tag = getattr(net, "tag", None)
        if self.tag not in ("conv2d", "conv3d", "ffdnet", "denoiser", "3d_denoiser"):
            raise NotImplementedError(f"unknown nonlinear_op tag {self.tag!r}")
        self.sigma_table = None
        self.fold_bn = fold_bn
        self.fast = None
        self._wkey = None
        self._refresh()

    def _weights_key(self):
        return tuple((t.data_ptr(), t._version) for t in list(self.net.parameters()) + list(self.net.buffers()))

    def _refresh(self):
        """(Re)build the BN-folded functional FFDNet whenever the module's tensors changed
        (e.g. load_state_dict after the engine was created)."""
        from .networks import FFDNet
        net = self.net
        key = (self._weights_key(), net.training)
        if key == self._wkey:
            return
        self._wkey = key
        self.fast = None
        self.wino = None
        self.tail_w = self.head_w = None
        self.plain_head_w = self.plain_tail_w = None
        from .networks import DnCNN
        from .networks.simplecnn import RealSNConv2d
        seq = None
        if isinstance(net, FFDNet) and not net.training and net.num_input_channels == 1 and self.fold_bn:
            seq = net.intermediate_dncnn.itermediate_dncnn
        elif isinstance(net, DnCNN) and not net.training and self.fold_bn and all(
                isinstance(mod, (torch.nn.Conv2d, RealSNConv2d, torch.nn.BatchNorm2d, torch.nn.ReLU)) for mod in net.dncnn):
            seq = net.dncnn                       # SimpleCNN / RealSN_SimpleCNN / DnCNN-17: conv [+BN] + ReLU blocks
        if seq is not None:
            mods = list(seq)
            layers, i = [], 0
            while i < len(mods):
                conv = mods[i]
                # RealSNConv2d in eval mode = conv2d with its stored, already normalised `weight` buffer
                # (networks/provable/model/conv_sn_chen.py:65-67): the same HIP kernels run it
                assert isinstance(conv, RealSNConv2d) or (isinstance(conv, torch.nn.Conv2d) and conv.bias is None)
                w, b = conv.weight.detach(), None
                i += 1
                if i < len(mods) and isinstance(mods[i], torch.nn.BatchNorm2d):
                    w, b = _fold_bn(w, mods[i])
                    i += 1
                relu = i < len(mods) and isinstance(mods[i], torch.nn.ReLU)
                if relu:
                    i += 1
                w = w.contiguous(memory_format=torch.channels_last) if self.channels_last else w.contiguous()
                layers.append((w, b, relu))
            self.fast = layers
            # 64->64 layers: Winograd on the fp32 matrix cores with bias+ReLU fused: F(4x4,3x3) (csrc/winograd44.hip) when the
            # launch has more than a wave of block tiles, F(2x2,3x3) (csrc/winograd.hip) below that - _hip.conv3x3_c64 picks
            # (the split-fp16 pack costs a host sync: made here, never inside a hipGraph capture, whenever the policy can pick that kernel)
            s16 = self.conv64 in ("fast", "s16")
            self.wino = [(_hip.pack_conv64_weights(w, s16=s16) if (self.winograd and self.channels_last and w.is_cuda
                                                                      and tuple(w.shape) == (64, 64, 3, 3)) else None)
                         for w, _, _ in layers]
            self.ranges = None
            self._stacks = {}
            self._wstacks = {}

            self.tail_w = self.head_w = None
            self.plain_head_w = self.plain_tail_w = None
            self.tail_w16 = self.plain_tail_w16 = self.head_w16 = None
            if (not isinstance(net, FFDNet) and self.fused_edges and self.channels_last and layers[0][0].is_cuda):
                # SimpleCNN-style stacks: 1 -> 64 (+ReLU) and 64 -> 1 edge layers as HIP stencils (csrc/ffdnet_edges.hip)
                if tuple(layers[0][0].shape) == (64, 1, 3, 3) and layers[0][1] is None:
                    self.plain_head_w = _hip.pack_c1_to_64_weights(layers[0][0])
                if tuple(layers[-1][0].shape) == (1, 64, 3, 3) and layers[-1][1] is None and not layers[-1][2]:
                    self.plain_tail_w = _hip.pack_c64_to_1_weights(layers[-1][0])
                    self.plain_tail_w16 = _hip.TailSplit16Weights(layers[-1][0])
                if (s16 and self.stack_kernel == "w16" and self.conv64 != "s16"
                        and self.plain_head_w is not None and self.plain_tail_w16 is not None):
                    for u in self.wino[1:-1]:                    # (the Winograd pack of the middle layers: here, never inside a capture)
                        if u is not None:
                            u.w16
            if (isinstance(net, FFDNet) and self.fused_edges and self.channels_last and layers[-1][1] is None
                    and not layers[-1][2] and tuple(layers[-1][0].shape) == (4, 64, 3, 3) and layers[-1][0].is_cuda):
                # last layer + upsamplefeatures as one HIP stencil kernel (csrc/ffdnet_edges.hip)
                self.tail_w = _hip.pack_tail_weights(layers[-1][0])
                self.tail_w16 = _hip.TailSplit16Weights(layers[-1][0])          # the same layer for an sp16 input (MFMA form)
                if layers[0][1] is None and layers[0][2] and tuple(layers[0][0].shape) == (64, 5, 3, 3):
                    # concatenate_input_noise_map + first conv + ReLU likewise
                    self.head_w = _hip.pack_head_weights(layers[0][0])
                    self.head_w16 = _hip.HeadSplit16Weights(layers[0][0])        # the same layer writing sp16 (f16 matrix-core form)

    def _run_stack(self, h, skip_last=False, skip_first=False, defer_last_epilogue=False, native_out=False):
        """defer_last_epilogue: leave the bias+ReLU of the last executed layer to the consumer (the fused
        FFDNet tail applies it while staging its input) and return (raw conv output, bias) - only when that
        layer runs on MIOpen; the HIP 64->64 kernels apply bias+ReLU in their own epilogue for free.
        native_out: the consumer reads the split-fp16 kernel's sp16 layout directly (the HIP tails do): do not convert back."""
        self._native_out = native_out
        if self.channels_last and not isinstance(h, _hip.Sp16):
            h = h.contiguous(memory_format=torch.channels_last)
        fused = self.fused_epilogue and h.is_cuda
        lo, hi = (1 if skip_first else 0), (len(self.fast) - 1 if skip_last else len(self.fast))
        idx = list(range(lo, hi))
        if defer_last_epilogue:
            last = idx[-1]
            w, b, relu = self.fast[last]
            if self.wino[last] is not None and h.is_cuda:
                return self._run_layers(h, idx, fused), None
            assert b is not None and relu
            return F.conv2d(self._run_layers(h, idx[:-1], fused), w, None, padding=1), b
        return self._run_layers(h, idx, fused)

    # a run shorter than this keeps its per-layer launches: the stack launch earns its keep over many layers on a cache-resident slice
    # (FFDNet's 13: 138.8 -> 144.8 frames/s); on SimpleCNN's two layers it costs 4 % (180.6 vs 188.7: agent-scope loads re-read the halo
    # overlap from memory instead of the L2, and there is no kernel boundary worth saving between two 100 us launches)
    STACK_MIN_LAYERS = 3

    def _stack_for(self, idx, device):
        """The Split16Stack of the run of layers idx (an H2D copy of its table: built by prepare(), never inside a hipGraph capture)."""
        st = self._stacks.get(idx[0])
        if st is None or st.n_layers != len(idx) or st.table.device != torch.device(device):
            st = self._stacks[idx[0]] = _hip.Split16Stack([(self.wino[i].s16, self.fast[i][1], self.fast[i][2]) for i in idx], device)
        return st

    def _wstack_for(self, idx, device):
        """The Wino16Stack of the run of layers idx (built by prepare(), never inside a hipGraph capture)."""
        st = self._wstacks.get(idx[0])
        if st is None or st.n_layers != len(idx) or st.table.device != torch.device(device):
            st = self._wstacks[idx[0]] = _hip.Wino16Stack([(self.wino[i].w16, self.fast[i][1], self.fast[i][2]) for i in idx], device)
        return st

    def _middle_run(self):
        """Indices of the run of 64->64 layers between the edge layers, when the whole run can take the split-fp16 kernel."""
        if self.fast is None or self.wino is None or len(self.fast) < 4 or self.conv64 not in ("fast", "s16"):
            return None
        idx = list(range(1, len(self.fast) - 1))
        return idx if all(self.wino[i] is not None for i in idx) else None

    def stack_slice(self, bsz, B, H, W, device):
        """Images per stack launch when an f-call of this shape goes slice by slice - first layer -> the run of 64->64 layers as ONE stack
        launch -> last layer (the branch of _run below; a calibrating call aside) - else None.  prepare() must have run."""
        if self.tag != "ffdnet" or self.fast is None or self.head_w is None or self.tail_w is None or torch.device(device).type != "cuda":
            return None
        run = self._middle_run()
        if (not self.stack or not self.slice_edges or run is None or len(run) < self.STACK_MIN_LAYERS or run[0] not in self._stacks
                or self._stacks[run[0]].n_layers != len(run) or not all(u is not None for u in self.wino[1:-1])
                or _hip.conv64_kernel_for(bsz * B, H // 2, W // 2, device, self.conv64) != "s16"):
            return None
        w16 = self.stack_kernel == "w16" and run[0] in self._wstacks and self._wstacks[run[0]].n_layers == len(run)
        st = self._wstacks[run[0]] if w16 else self._stacks[run[0]]
        return self.stack_per_launch or _hip.split16_stack_per_launch(bsz * B, H // 2, W // 2,
                                                                      cus=torch.cuda.get_device_properties(device).multi_processor_count, tile=st.TILE)

    def stack_timed_out(self):
        """(host sync) whether a wait inside a stack launch gave up since the last call: its results are invalid."""
        return any([st.timed_out() for st in list(self._stacks.values()) + list(self._wstacks.values())])    # (every stack: the words are rearmed)

    def _run_layers(self, h, idx, fused):
        if (self.stack and isinstance(h, _hip.Sp16) and len(idx) >= self.STACK_MIN_LAYERS and self._native_out and not self._calibrating
                and idx[0] in self._stacks and self._stacks[idx[0]].n_layers == len(idx) and (h.rng is None) == (self.ranges is None)):
            # the whole run in one launch (per slice of the batch): sp16 in, sp16 out, ranges of the run = slots idx[0] .. idx[-1] + 1
            self.stack_launches += 1
            return _hip.conv3x3_c64_split16_stack(h, self._stack_for(idx, h.t.device), None if self.ranges is None else self.ranges[idx[0]:idx[-1] + 2],
                                                  per_launch=self.stack_per_launch, check=False)
        for pos, i in enumerate(idx):
            w, b, relu = self.fast[i]
            nxt = idx[pos + 1] if pos + 1 < len(idx) else None
            chain = nxt is not None and self.wino[nxt] is not None          # the next layer is a 64->64 layer too: keep the kernel's own layout
            native = isinstance(h, (_hip.Sp16, _hip.Blk32))
            if self.wino[i] is not None and (native or (h.is_cuda and h.is_contiguous(memory_format=torch.channels_last))):
                # 64->64 layer on one of the three HIP kernels; between two such layers the activation stays in the kernel's own layout
                # (sp16 for the split-fp16 direct convolution, blk32 for F(4x4,3x3)); the layer's kind is fixed by the FIRST layer of a run
                kind = ("s16" if isinstance(h, _hip.Sp16) else "f44" if isinstance(h, _hip.Blk32) else
                        _hip.conv64_kernel_for(h.shape[0], h.shape[2], h.shape[3], h.device, self._policy))
                if kind == "s16":
                    if not isinstance(h, _hip.Sp16):
                        if self._calibrating:
                            _hip.absmax(h, self._slot(i))
                            self._measured = True
                        h = _hip.to_split16(h, rng=self._slot(i))
                    sp_out = chain or (nxt is None and self._native_out)
                    if sp_out and self._calibrating:               # measure max |output| of this layer, then write it with that range
                        _hip.conv3x3_c64_split16(h, self.wino[i].s16, b, relu, track=self._slot(i + 1))
                        self._measured = True
                    h = _hip.conv3x3_c64_split16(h, self.wino[i].s16, b, relu, out_f32=not sp_out, out_rng=self._slot(i + 1) if sp_out else None)
                elif kind == "f44":
                    h = _hip.conv3x3_c64_winograd44(h, self.wino[i].f44, b, relu, out_blk=bool(self.blk32 and chain))
                else:
                    h = _hip.conv3x3_c64_winograd(h, self.wino[i].f22, b, relu)
            elif fused and b is not None:
                # Conv-BN-ReLU = MIOpen conv with folded weights + ONE in-place bias+ReLU pass (HIP)
                h = _hip.bias_relu_(F.conv2d(h, w, None, padding=1), b, relu)
            else:
                h = F.conv2d(h, w, b, padding=1)
                if relu:
                    h = F.relu_(h)
        return h

    def _slot(self, i):
        """Range slots (one per image) of the input of layer i (= the output of layer i - 1), or None under act_range="fixed"."""
        return None if self.ranges is None else self.ranges[i]

    def _alloc_ranges(self, n_img, device):
        if self.fast is not None and self.act_range == "data" and (self.ranges is None or self.ranges.device != torch.device(device)
                                                                    or self.ranges.shape[1] != n_img):
            # kept across calls of a shape: a captured hipGraph carries this tensor's address in its conv nodes (a new batch size is a
            # new graph, captured after an eager call)
            self.ranges = torch.zeros((len(self.fast) + 1, n_img), dtype=torch.float32, device=device)
            self._stale = True

    def prepare(self, n_calls, device, n_img=None):
        self._refresh()
        if n_img is not None:
            self._alloc_ranges(n_img, device)
        run = self._middle_run() if (self.stack and torch.device(device).type == "cuda") else None
        if run is not None and len(run) >= self.STACK_MIN_LAYERS and self.wino[run[0]].s16.packed.is_cuda:
            self._stack_for(run, device)
            if self.stack_kernel == "w16" and self.tag == "ffdnet" and self.head_w16 is not None and self.tail_w16 is not None:
                self._wstack_for(run, device)
        if self.tag == "ffdnet":
            t = self.sigma_table
            if t is None or t.numel() < n_calls or t.device != torch.device(device):
                # kept across calls: a captured hipGraph carries this tensor's address in its FFDNet-head nodes
                self.sigma_table = torch.from_numpy(sigma_schedule(n_calls)).to(device)

    def run(self, z1, call, calibrate=False):
        return _exhaust(self.run_steps(z1, call, calibrate))

    def run_steps(self, z1, call, calibrate=False):
        """calibrate: a new input (the engine: the first f-call of a reconstruction) - the ranges of the split-fp16 activations measured
        last are stale.  They are measured by the next call that takes the split-fp16 path (this one, unless the policy runs its first
        f-calls on another kernel) and kept from then on."""
        bsz, B, H, W = z1.shape
        x = z1.view(bsz * B, 1, H, W)
        self._policy = "f22" if (self.f22_calls is not None and call < self.f22_calls) else self.conv64
        if x.is_cuda:
            self._alloc_ranges(bsz * B, x.device)
        if calibrate:
            self._stale = True
        self._calibrating = cal = self._stale and self.ranges is not None and x.is_cuda
        self._measured = False                                      # set by the first measuring launch of this call
        if cal:
            self.ranges.zero_()
        try:
            out = yield from self._run_gen(z1, x, call, cal)
            if self._measured:                                      # (only a call that ran to its end has measured every layer)
                self._stale = False
            return out
        finally:
            self._calibrating = False

    def _run_gen(self, z1, x, call, cal):
        """(generator) the f-call's denoiser; yields "stack" behind every stack launch - where a grouped reconstruction switches to its other
        half (DEQSCIEngine._reconstruct_grouped) - and returns (output, is_noise)."""
        bsz, B, H, W = z1.shape
        if self.tag == "ffdnet":
            sig = self.sigma_table[call:call + 1].expand(bsz * B)
            if self.fast is not None:
                # a run of 64->64 layers on the split-fp16 kernel: the head writes its sp16 layout, the tail reads it - no conversion pass
                sp = (x.is_cuda and self.head_w is not None and self.tail_w is not None and all(u is not None for u in self.wino[1:-1])
                      and _hip.conv64_kernel_for(bsz * B, H // 2, W // 2, x.device, self._policy) == "s16")
                run = self._middle_run()
                if (sp and not cal and self.stack and self.slice_edges and run is not None and len(run) >= self.STACK_MIN_LAYERS
                        and run[0] in self._stacks and self._stacks[run[0]].n_layers == len(run)):
                    # slice by slice: first layer -> the run of 64->64 layers as one stack launch -> last layer, each handing its output to the
                    # next through the Infinity Cache (a slice's activation is at most 128 MiB: _hip.split16_stack_per_launch)
                    n, sg = bsz * B, self.sigma_table[call:call + 1]
                    w16 = self.stack_kernel == "w16" and run[0] in self._wstacks and self._wstacks[run[0]].n_layers == len(run)
                    st = self._wstacks[run[0]] if w16 else self._stacks[run[0]]
                    per = self.stack_per_launch or _hip.split16_stack_per_launch(n, H // 2, W // 2,
                                                                                  cus=torch.cuda.get_device_properties(x.device).multi_processor_count,
                                                                                  tile=st.TILE)
                    bufs, hbuf = st.state(min(per, n), H // 2, W // 2), st.head_buffer(min(per, n), H // 2, W // 2)
                    rows = None if self.ranges is None else self.ranges[run[0]:run[-1] + 2]
                    out = torch.empty((n, 1, H, W), dtype=torch.float32, device=x.device)
                    for a in range(0, n, per):
                        m = min(per, n - a)
                        hin = hbuf if m == hbuf.n else st.act(hbuf.t[:m], m, hbuf.H, hbuf.W)
                        in_rng = None if self.ranges is None else self._slot(0)[a:a + m]
                        out_rng = None if self.ranges is None else self._slot(1)[a:a + m]
                        if w16:
                            hs = _hip.ffdnet_head_p32(x[a:a + m], self.head_w16, sg, out=hin, in_rng=in_rng, out_rng=out_rng)
                            if self.gate is not None:
                                self.gate.acquire()
                            ys = _hip.conv3x3_c64_wino16_stack(hs, st, rows, per_launch=m, rng_offset=a, out_bufs=bufs, check=False)
                            if self.gate is not None:
                                self.gate.release()
                                yield "stack"
                            _hip.ffdnet_tail_p32(ys, self.tail_w16, out=out[a:a + m])
                        else:
                            hs = _hip.ffdnet_head_split16(x[a:a + m], self.head_w16, sg, out=hin, in_rng=in_rng, out_rng=out_rng)
                            if self.gate is not None:
                                self.gate.acquire()
                            ys = _hip.conv3x3_c64_split16_stack(hs, st, rows, per_launch=m, rng_offset=a, out_bufs=bufs, check=False)
                            if self.gate is not None:
                                self.gate.release()
                                yield "stack"
                            _hip.tail_split16(ys, self.tail_w16, out=out[a:a + m])
                    self.stack_launches += 1
                    self.last_path = ("w16" if w16 else "s16") + " stack launch"
                    return out.reshape(bsz, B, H, W), True
                if (sp and not cal and not self.stack and self.per_layer_w16 and self.stack_kernel == "w16" and self.ranges is not None
                        and run is not None and self.head_w16 is not None and self.tail_w16 is not None and run[0] in self._wstacks):
                    # behind a stack time-out: the same kernel, the same ranges, one launch per layer - the stack launch's bits
                    sg = self.sigma_table[call:call + 1]
                    h = _hip.ffdnet_head_p32(x, self.head_w16, sg, in_rng=self._slot(0), out_rng=self._slot(1))
                    for i in run:
                        h = _hip.conv3x3_c64_wino16(h, self.wino[i].w16, self.fast[i][1], self.fast[i][2], out_rng=self._slot(i + 1))
                    self.last_path = "w16 per layer (behind a stack time-out)"
                    return _hip.ffdnet_tail_p32(h, self.tail_w16).reshape(bsz, B, H, W), True
                self.last_path = "per layer"
                if self.head_w is not None and x.is_cuda:
                    sg = self.sigma_table[call:call + 1]
                    if sp:
                        if cal:                                # max |image| (the head adds sigma itself), then max |its own output|
                            self._measured = True
                            _hip.absmax(x, self._slot(0))
                            _hip.ffdnet_head_split16(x, self.head_w16, sg, in_rng=self._slot(0), out_exp=0, track=self._slot(1))
                        h = _hip.ffdnet_head_split16(x, self.head_w16, sg, in_rng=self._slot(0), out_rng=self._slot(1))
                    else:
                        h = _hip.ffdnet_head(x, self.head_w, sg)
                    first_done = True
                else:
                    h = torch.cat((sig.view(-1, 1, 1, 1).expand(bsz * B, 1, H // 2, W // 2), F.pixel_unshuffle(x, 2)), 1)
                    first_done = False
                if self.tail_w is not None and h.is_cuda:
                    defer = self.fused_epilogue and self.fast[-2][1] is not None and self.fast[-2][2]
                    if defer:
                        raw, b = self._run_stack(h, skip_last=True, skip_first=first_done, defer_last_epilogue=True, native_out=sp)
                    else:
                        raw, b = self._run_stack(h, skip_last=True, skip_first=first_done, native_out=sp), None
                    out = _hip.tail_split16(raw, self.tail_w16) if isinstance(raw, _hip.Sp16) else _hip.ffdnet_tail(raw, self.tail_w, in_bias=b)
                else:
                    out = F.pixel_shuffle(self._run_stack(h, skip_first=first_done), 2)
            else:
                out = self.net(x, sig)
            return out.reshape(bsz, B, H, W), True
        if self.tag == "denoiser":
            if self.fast is not None:
                if x.is_cuda and (self.plain_head_w is not None or self.plain_tail_w is not None):
                    first = self.plain_head_w is not None
                    sp = (first and self.plain_tail_w is not None and all(u is not None for u in self.wino[1:-1])
                          and _hip.conv64_kernel_for(bsz * B, H, W, x.device, self._policy) == "s16")
                    if first and sp and cal:
                        self._measured = True
                        _hip.conv3x3_c1_to_64(x, self.plain_head_w, relu=self.fast[0][2], sp16=True, out_exp=0, track=self._slot(1))
                    if (first and sp and not cal and self.stack_kernel == "w16" and self.plain_tail_w16 is not None and self.conv64 != "s16"
                            and all(u is not None for u in self.wino[1:-1]) and self.ranges is not None):
                        # the 64->64 layers on the split-fp16 Winograd kernel (csrc/conv_w16.hip: a third fewer matrix-core products), one
                        # launch per layer (two layers: no run worth a stack launch), p32 activations from the first layer to the last; the
                        # ranges are the ones the first f-call measured on the direct kernels below.  conv64="s16" means the direct kernel
                        # (ADVICE r5).  (A long DnCNN-style run - a user plugin; test_plugin_stack_dncnn17_follows_the_data_scale - takes this
                        # branch too: per-layer Winograd launches, a third fewer matrix-core products than its one direct stack launch.)
                        h = _hip.conv3x3_c1_to_64(x, self.plain_head_w, relu=self.fast[0][2], p32=True, out_rng=self._slot(1))
                        for i in range(1, len(self.fast) - 1):
                            h = _hip.conv3x3_c64_wino16(h, self.wino[i].w16, self.fast[i][1], self.fast[i][2], out_rng=self._slot(i + 1))
                        return _hip.ffdnet_tail_p32(h, self.plain_tail_w16).reshape(bsz, B, H, W), True
                    h = _hip.conv3x3_c1_to_64(x, self.plain_head_w, relu=self.fast[0][2], sp16=sp, out_rng=self._slot(1) if sp else None) if first else x
                    if self.plain_tail_w is not None:
                        h = self._run_stack(h, skip_first=first, skip_last=True, native_out=sp)
                        out = (_hip.tail_split16(h, self.plain_tail_w16) if isinstance(h, _hip.Sp16) else
                               _hip.conv3x3_c64_to_1(h.contiguous(memory_format=torch.channels_last), self.plain_tail_w))
                    else:
                        out = self._run_stack(h, skip_first=first)
                    return out.reshape(bsz, B, H, W), True
                return self._run_stack(x).reshape(bsz, B, H, W), True
            return self.net(x).reshape(bsz, B, H, W), True
        if self.tag == "conv2d":
            return self.net(x).reshape(bsz, B, H, W), False
        x5 = z1.view(bsz, 1, B, H, W)
        return self.net(x5).reshape(bsz, B, H, W), self.tag == "3d_denoiser"


class _StackTimeout(Exception):
    """A wait inside a stack launch gave up (seen behind the first stack f-call of an eager reconstruction)."""


class _StackGate:
    """One stack launch on the device at a time, whatever stream it is on.  The workgroups of a stack launch wait for one another, so all of
    them have to be resident; two such launches dispatched side by side from two streams can each hold part of the CUs and starve the other
    until the waits give up.  Every stack launch of a grouped reconstruction therefore waits (on the device: a stream-side event wait, the
    host goes on) for the stack launch issued before it - a chain in issue order, no cycle."""

    def __init__(self):
        self.ev = None

    def acquire(self):
        if self.ev is not None:
            torch.cuda.current_stream().wait_event(self.ev)

    def release(self):
        self.ev = torch.cuda.Event()
        self.ev.record()


class DEQSCIEngine:
    GRAPH_AUTO_PIXELS = 4 * 256 * 256
    STACK_RETRY_CALLS = 16        # reconstructions on per-layer launches after a stack launch timed out, before the stack launch is tried again

    def __init__(self, denoiser, iterator="anderson", m=5, beta=1.0, lam=1e-2, max_iter=180, tol=1e-5,
                 fold_bn=True, extra_call=False, poll_residual=True, channels_last=None, fused_epilogue=True,
                 fused_edges=True, winograd=True, use_graph="auto", conv64="auto", conv64_f22_calls=None, act_range="data", blk32=True, anderson_arith="reference",
                 stack=True, stack_kernel="w16", groups=1):
        if iterator not in ("anderson", "picard"):
            raise ValueError(iterator)
        if not (groups == "auto" or (isinstance(groups, int) and 1 <= groups <= 2)):
            raise ValueError(f"groups={groups!r}: expected 'auto', 1 or 2")
        # groups: 2 (or "auto") = a batch of at least two stack slices (FFDNet: 2 x 32 images = 8 measurements of 256 x 256 x 8) is reconstructed
        # as TWO independent half batches, each on its own stream, issued alternately up to every stack launch (_reconstruct_grouped).
        # Measurements are independent problems and every kernel of the path is per measurement, so the result is bit-identical; what
        # changes is what the device overlaps: a stack launch holds every CU, but the dozen short kernels around it (first / last layer, K4,
        # the Gram kernels, the 6 x 6 solve, K7+K3 - a seventh of an f-call) of one half run beside the other half's.  MEASURED, and NOT the
        # default (1 = one stream): the short kernels are HBM- or latency-bound and run at the clock the power-capped stack launch leaves
        # behind; side by side they gain 50 us per f-call, the stack launches lose 13-36 us each waiting for CUs the other half's short
        # kernels still hold, and the outcome is between -4 % and +1.3 % from box to box and configuration to configuration
        # (profiles/r06_groups_*.txt).
        self.groups = groups
        self._ctor = dict(iterator=iterator, m=m, beta=beta, lam=lam, max_iter=max_iter, tol=tol, fold_bn=fold_bn, extra_call=extra_call,
                          channels_last=channels_last, fused_epilogue=fused_epilogue, fused_edges=fused_edges, winograd=winograd, conv64=conv64,
                          conv64_f22_calls=conv64_f22_calls, act_range=act_range, blk32=blk32, anderson_arith=anderson_arith, stack=stack,
                          stack_kernel=stack_kernel)
        self._net = denoiser
        self._kids, self._gate, self._grouped = None, None, False
        if anderson_arith not in ("float64", "reference"):
            raise ValueError(f"anderson_arith={anderson_arith!r}: expected 'reference' or 'float64'")
        # How alpha is computed.  "reference" (the default of every entry point: this class, the drop-in DEQFixedPoint / andersonexp, the CLI,
        # bench.py): the reference's own arithmetic for that step, solvers/new_equilibrium_utils_yaping.py:177-180 - G G^T as ONE fp32 torch.bmm
        # over the N = H W B elements, fp32 LU - by the build's own kernels: the new Gram row in the summation ORDER of that GEMM on the CPU
        # behind tests/golden (MKL sgemm on the build host's AVX-512 CPU: 16 interleaved FMA chains per entry, csrc/anderson.hip; no GEMM
        # library, capturable), the system formed and factorised in fp32 by K6 as sgesv does.  It exists because that order is not neutral on
        # BASELINE config 2: a 2^15-step chain absorbs the many small products of the heavy-tailed residuals, the diagonal of the Gram matrix
        # comes out 3-7e-6 too small, and that bias is what puts the reference's 180-iteration ensemble means where they are (DESIGN section
        # 5, "Config 2").  Another BLAS build or CPU orders the sum differently: "the reference's fp32 Gram" is a property of the machine the
        # reference ran on, and this is the build host's.  "float64": the Gram row accumulated in float64 from the fp32 block partials of K4,
        # the bordered system solved in float64 (K5+K6) - alpha to ~1e-8, 2-4 % faster at eight measurements per call, ~10 % at one; every
        # well-conditioned configuration is indifferent.  (Round 4's rocBLAS torch.bmm form lives in tools/config2_anderson_arith.py.)
        self.anderson_arith = anderson_arith
        if conv64 not in ("auto", "fast", "fast32", "f22", "f44", "s16"):
            raise ValueError(f"conv64={conv64!r}: expected 'auto', 'fast', 'fast32', 'f22', 'f44' or 's16'")
        # Which kernel runs the 64->64 layers (_hip.conv64_kernel_for): "fast" (= "auto") the faster of the split-fp16 direct convolution
        # on the f16 matrix cores and Winograd F(2x2,3x3) on the f32 ones, per launch; "fast32" fp32 MFMA arithmetic only (F(4x4,3x3) /
        # F(2x2,3x3)); "f22" / "f44" / "s16" one kernel always.  `conv64_f22_calls=K` additionally runs the first K f-calls on F(2x2,3x3).
        # What the choice does to the result was measured where it can matter - FFDNet under Anderson beyond ~30 iterations is chaotic
        #  - as 25-start ensembles over the six traffic measurements (tools/config2_ensemble.py,
        # profiles/r03_config2_ensembles.json; mean PSNR, standard error 0.004): reference 21.434 +- 0.004 (as it is 21.439 +- 0.005); F(2x2,3x3)
        # 21.420; split-fp16 21.428; F(2x2,3x3) for 40 f-calls then F(4x4,3x3) 21.417; MIOpen's direct fp32 convolution 21.410;
        # F(4x4,3x3) throughout 21.395.  Single measurements move by up to 0.09 dB under ANY change of arithmetic (the reference's own two
        # Gram variants: RMS 0.065 dB), so only the pooled mean separates kernels; it puts split-fp16 and F(2x2,3x3) together, nearest
        # the reference, which is why "auto" uses exactly those two.  Well-conditioned configurations (SimpleCNN, Picard, <= 30
        # iterations) agree to 1e-5 under every policy.
        self.conv64 = conv64
        self.conv64_policy = "fast" if conv64 == "auto" else conv64
        self.conv64_f22_calls = None if conv64_f22_calls is None else int(conv64_f22_calls)
        # act_range: "data" (default) = the power-of-two scales of the split-fp16 activations follow the data, measured on the device at
        # the first f-call of every reconstruction (fp32, the reference's arithmetic at equilibrium_solvers_yaping.py:397-420, is
        # scale-free; fp16 pieces are not); "fixed" = 2^8 throughout, the round-3 behaviour (activations of a few units).
        self.den = _Denoiser(denoiser, fold_bn=fold_bn, channels_last=channels_last, fused_epilogue=fused_epilogue,
                             fused_edges=fused_edges, winograd=winograd, conv64=self.conv64_policy, act_range=act_range, blk32=blk32, stack=stack,
                             stack_kernel=stack_kernel)
        self.den.f22_calls = self.conv64_f22_calls
        self.iterator = iterator
        self.m, self.beta, self.lam = int(m), float(beta), float(lam)
        self.max_iter, self.tol = int(max_iter), float(tol)
        self.extra_call = extra_call
        self.poll_residual = poll_residual
        # True / False / "auto": replay a captured hipGraph when one reconstruction is at most GRAPH_AUTO_PIXELS
        # measurement-pixels (4 measurements of 256x256), i.e. when launch gaps are a visible share of the run
        self.use_graph = use_graph
        self._eager = True                # (False while a hipGraph capture records the launches: no host sync then)
        self._early_check = True          # (False in the halves of a grouped reconstruction: the parent looks at the time-out words once, at the end)
        self._stack_wanted, self._stack_off_for, self.stack_timeouts_total = bool(stack), 0, 0
        self._ws = {}
        self._graph = None
        self.last_info = None

    # ------------------------------------------------------------------ buffers
    def _workspace(self, bsz, H, W, B, device):
        key = (bsz, H, W, B, str(device), self.iterator, self.m, self.max_iter)
        ws = self._ws.get(key)
        if ws is None:
            m = self.m if self.iterator == "anderson" else 1
            rows = self.max_iter + 2
            ws = _hip.AndersonWorkspace(bsz, H * W * B, m, device, res_rows=rows)
            ws.xbuf = [torch.empty((bsz, B, H, W), device=device, dtype=torch.float32) for _ in range(2)]
            ws.z1 = torch.empty((bsz, B, H, W), device=device, dtype=torch.float32)
            ws.host_res = torch.full((rows, 1 + bsz), float('inf'), dtype=torch.float32).pin_memory()   # unfinished copy != converged
            self._ws = {key: ws}          # one live shape at a time: history is bsz*m*N*8 bytes
        return ws

    # ------------------------------------------------------------------ one f-call = GAP -> denoise -> store -> solve
    def _store_solve(self, ws, x_in, call, slot, n_filled, n_solve, x_next, eps, res_row):
        return _exhaust(self._store_solve_steps(ws, x_in, call, slot, n_filled, n_solve, x_next, eps, res_row))

    def _store_solve_steps(self, ws, x_in, call, slot, n_filled, n_solve, x_next, eps, res_row):
        # (the first f-call measures the activations' ranges; only a half of a grouped reconstruction - a gate is set - needs the generator form)
        if self.den.gate is not None:
            out, is_noise = yield from self.den.run_steps(ws.z1, call, calibrate=(call == 0))
        else:
            out, is_noise = self.den.run(ws.z1, call, calibrate=(call == 0))
        if call == 1 and self._eager and self._early_check and self.den.stack and self.den.stack_launches and self.den.stack_timed_out():
            # the FIRST stack launch of the reconstruction: a wait that gave up (its workgroups were not all resident: somebody else holds CUs)
            # is seen here, one f-call in - not after 180 f-calls on invalid data.  One host sync per reconstruction (~0.1 ms of queue refill).
            raise _StackTimeout()
        out = _hip.f32c(out)
        ref = self.anderson_arith == "reference"               # the reference's fp32 Gram formed by K4 + K5 themselves (no GEMM library)
        if is_noise:
            _hip.residual_store(ws, ws.z1, out, x_in, slot, n_filled, x_next, ref=ref)
        else:
            _hip.residual_store(ws, out, None, x_in, slot, n_filled, x_next, ref=ref)
        _hip.anderson_solve(ws, slot, n_filled, n_solve, self.lam, eps, res_row, ref=ref)    # (+ lam I and the fp32 LU of :178-180 in K6)

    def _poll(self, ws, row):
        ws.host_res[row].copy_(ws.res[row], non_blocking=True)
        ev = torch.cuda.Event()
        ev.record()
        return ev

    # ------------------------------------------------------------------ public
    @torch.no_grad()
    def reconstruct(self, y, Phi, Phi_sum=None, initial_point=None):
        """y (bsz,H,W), Phi (bsz|1,H,W,B) or (H,W,B) fp32 on the GPU -> reconstruction (bsz,H,W,B).
        `self.last_info` holds res (whole batch), per-sample res, iterations and f-call count."""
        if not (isinstance(y, torch.Tensor) and y.is_cuda):
            raise _hip.DeqsciHipError("DEQSCIEngine.reconstruct needs GPU tensors; there is no CPU path")
        with torch.cuda.device(y.device):          # events, stream sync and launches all on y's device
            if self._stack_off_for > 0:            # (a stack launch timed out a while ago: per-layer launches for STACK_RETRY_CALLS calls, then try again)
                self._stack_off_for -= 1
                if self._stack_off_for == 0 and self._stack_wanted:
                    self.den.stack, self.den.per_layer_w16, self._graph = True, False, None
            timed_out = 0
            try:
                rec = self._reconstruct(y, Phi, Phi_sum, initial_point)
                if self.den.stack and self.den.stack_launches and self._stack_timed_out():
                    raise _StackTimeout()          # (a replayed hipGraph, a grouped call, or a wait that gave up later than the first stack f-call)
            except _StackTimeout:
                # a stack launch waits for its own workgroups only - all resident when the device is ours.  A wait that gave up means it is
                # not (another process holds CUs): that result is invalid; redo it with a launch per layer, and stay there for a while
                import warnings
                warnings.warn("deqsci_amd: a wait inside a split-fp16 stack launch timed out (the device's CUs are shared with other work); "
                              f"redoing this call with one launch per layer and keeping that for the next {self.STACK_RETRY_CALLS} calls of this engine",
                              RuntimeWarning)
                self._stack_timed_out()            # (rearm every stack's words)
                self.den.stack = False
                self.den.per_layer_w16 = True      # (FFDNet under the Winograd kernel: per-layer launches of THAT kernel, the stack launch's bits)
                self.den.stack_launches = 0
                self._graph = None
                self._stack_off_for = self.STACK_RETRY_CALLS
                self.stack_timeouts_total += 1
                timed_out = 1
                rec = self._reconstruct(y, Phi, Phi_sum, initial_point)
            self.last_info["stack_launches"], self.den.stack_launches = self.den.stack_launches, 0
            self.last_info["stack_timeouts"] = timed_out     # 1: a stack launch of THIS call timed out (the result below is the per-layer redo)
            # what the run of 64->64 layers of the LAST f-call went out as (the kernel actually used, for every call)
            self.last_info["denoiser_path"] = (self._kids[0].den.last_path if (self._grouped and self._kids) else self.den.last_path)
            fallback = None
            if self.conv64 == "auto" and not math.isfinite(self.last_info["res"]) and bool(torch.isfinite(y).all()):
                # conv64="auto" promises the reference's fp32 range: a non-finite residual under the split-fp16 layers (already warned
                # about) is redone on the fp32 MFMA kernels - THIS call only (eagerly; a captured hipGraph and the policy of later,
                # unrelated inputs are left alone; every rank of a sharded job decides for its own call) - and last_info says so.  A run
                # that diverges by itself comes back non-finite again and is returned as it is.
                saved = (self.conv64_policy, self.den.conv64, self.use_graph)
                self.conv64_policy = self.den.conv64 = self.den._policy = "fast32"
                self.use_graph = False
                try:
                    rec = self._reconstruct(y, Phi, Phi_sum, initial_point)
                finally:
                    self.conv64_policy, self.den.conv64, self.use_graph = saved
                    self.den._policy = saved[1]
                fallback = "fast32"
            self.last_info["conv64_fallback"] = fallback
            # what the first f-call measured: max |activation| in front of every layer of the denoiser's stack, here the maximum over the
            # batch's images (the kernels use one range per image: den.ranges; 0 where no split-fp16 layer ran)
            rng = self._all_ranges()
            self.last_info["act_ranges"] = None if (rng is None or fallback) else rng.max(dim=1).values.tolist()
            return rec

    def _stack_timed_out(self):
        """(host sync) whether a wait inside a stack launch of this engine or of the halves of a grouped call gave up; rearms the words."""
        dens = [self.den] + [k.den for k in (self._kids or [])]
        return any([d.stack_timed_out() for d in dens])

    def _all_ranges(self):
        """(layers + 1, images) range slots of the last call (a grouped call: its halves' side by side)."""
        if self._grouped and self._kids:
            parts = [k.den.ranges for k in self._kids]
            return None if any(p is None for p in parts) else torch.cat(parts, dim=1)
        return self.den.ranges

    def _reconstruct(self, y, Phi, Phi_sum, initial_point):
        y = _hip.f32c(y)
        bsz, H, W = y.shape
        Phi4 = _hip.f32c(Phi if Phi.dim() == 4 else Phi.unsqueeze(0))
        B = Phi4.shape[3]
        if Phi4.shape[0] not in (1, bsz) or tuple(Phi4.shape[1:3]) != (H, W):
            raise _hip.DeqsciHipError(f"Phi {tuple(Phi.shape)} does not match y {tuple(y.shape)}")
        if Phi_sum is not None and Phi_sum.numel() == Phi4.shape[0] * H * W:
            Phi_sum = _hip.f32c(Phi_sum).view(Phi4.shape[0], H, W)
        else:
            Phi_sum = None
        if initial_point is not None:
            initial_point = _hip.f32c(initial_point)
        graph = self.use_graph if self.use_graph != "auto" else bsz * H * W <= self.GRAPH_AUTO_PIXELS
        self._grouped = False
        plan = None if graph else self._group_plan(bsz, H, W, B, y.device)
        if plan is not None:
            rec = self._reconstruct_grouped(plan, y, Phi4, Phi_sum, initial_point)
            if rec is not None:
                return rec
        ws = self._workspace(bsz, H, W, B, y.device)
        self.den.prepare(self.max_iter + 4, y.device, n_img=bsz * B)
        if graph:
            rec = self._replay(ws, y, Phi4, Phi_sum, initial_point)
            if rec is not None:
                return rec
        rec, call, last, res_row = self._enqueue(ws, y, Phi4, Phi_sum, initial_point, self.poll_residual)
        torch.cuda.current_stream().synchronize()
        r = ws.host_res[res_row]
        self.last_info = {"res": float(r[0]), "res_per_sample": r[1:].tolist(), "iterations": last,
                          "f_calls": call, "iterator": self.iterator, "graph": False}
        self._warn_if_not_finite()
        return rec

    # ------------------------------------------------------------------ two half batches on two streams
    def _group_plan(self, bsz, H, W, B, device):
        """[(lo, hi), (lo, hi)]: the measurements of the two halves of a grouped reconstruction, or None (groups=1; a batch of less than two
        stack slices; a denoiser whose f-call is not the sliced stack path).  A half is a whole number of stack slices."""
        if self.groups == 1 or bsz < 2:
            return None
        self.den.prepare(self.max_iter + 4, device)
        per = self.den.stack_slice(bsz, B, H, W, device)
        if per is None or per % B or bsz * B < 2 * per:
            return None
        mps = per // B                                        # measurements per stack slice
        first = (-(-(-(-bsz // mps)) // 2)) * mps             # ceil(slices / 2) slices
        return [(0, first), (first, bsz)]

    def _make_kids(self, n, device):
        if self._kids is None or len(self._kids) != n or self._kids[0].stream.device != torch.device(device):
            self._gate = _StackGate()
            self._kids = []
            for _ in range(n):
                kid = DEQSCIEngine(self._net, use_graph=False, groups=1, poll_residual=False, **self._ctor)
                kid.den.gate = self._gate
                kid._early_check = False
                kid.stream = torch.cuda.Stream(device=device)
                self._kids.append(kid)
        for kid in self._kids:                                # (what callers set on an engine after building it)
            for name in ("m", "beta", "lam", "max_iter", "tol", "extra_call", "iterator", "anderson_arith", "conv64", "conv64_policy", "conv64_f22_calls"):
                setattr(kid, name, getattr(self, name))
            for name in ("stack", "stack_per_launch", "slice_edges", "f22_calls", "conv64", "stack_kernel", "blk32"):
                setattr(kid.den, name, getattr(self.den, name))
        return self._kids

    def _reconstruct_grouped(self, plan, y, Phi4, Phi_sum, initial_point):
        """The batch as independent half batches (plan), each on its own stream with its own workspace and activation buffers, their
        f-calls issued alternately; the stack launches of the two streams are chained by events (_StackGate).  Bit-identical to the
        one-stream path (every kernel of the path is per measurement).  The tolerance test of new_equilibrium_utils_yaping.py:184-186 is
        over the WHOLE batch: the halves run max_iter iterations, and if any half's own residual ever fell below tol (then, and only then,
        the whole batch's might have: sqrt(sum g) / (eps + sqrt(sum f)) >= tol whenever that holds for every half) the call is redone on
        the one-stream path, which stops where the reference stops -> None.  Never on the reference's data."""
        bsz, H, W = y.shape
        B = Phi4.shape[3]
        dev = y.device
        kids = self._make_kids(len(plan), dev)
        main = torch.cuda.current_stream()
        gens, wss = [], []
        try:
            for kid, (lo, hi) in zip(kids, plan):
                kid.stream.wait_stream(main)
                with torch.cuda.stream(kid.stream):
                    ws = kid._workspace(hi - lo, H, W, B, dev)
                    # K4 and the reference Gram's first pass stay TWO launches here: the blocks of the fused launch wait for one another inside
                    # it (a bounded look-back), and a stack launch of the other half that takes the CUs in the middle of its dispatch leaves
                    # them spinning on blocks that cannot start - measured: stack launches 1017 -> 1053 us, the grouped step slower than one stream
                    ws.ref_fusable = False
                    kid.den.prepare(self.max_iter + 4, dev, n_img=(hi - lo) * B)
                    kid.den.stack_launches = 0
                    wss.append(ws)
                    gens.append(kid._enqueue_steps(ws, y[lo:hi], Phi4 if Phi4.shape[0] == 1 else Phi4[lo:hi],
                                                   None if Phi_sum is None else (Phi_sum if Phi_sum.shape[0] == 1 else Phi_sum[lo:hi]),
                                                   None if initial_point is None else initial_point[lo:hi], False))
            # One ROUND = every half issued up to (and including) its next stack launch; then every half that launched one makes its stream
            # wait for the LAST stack launch of the round before it goes on.  Without that wait the order on the device is a race at every
            # stack's end - the finished half's own short kernels (next in ITS stream) against the other half's stack launch (released by an
            # event) - and it locks into the pattern stack A | A's short kernels | stack B | B's short kernels: nothing overlaps, and the
            # halved launches make it SLOWER than one stream (443 against 426 ms per step).  With it: stack A | stack B | the short
            # kernels of both halves side by side | stack A | ...
            done = [None] * len(kids)
            alive = list(range(len(kids)))
            while alive:
                stacked = []
                for i in list(alive):
                    with torch.cuda.stream(kids[i].stream):
                        try:
                            while next(gens[i]) != "stack":
                                pass
                            stacked.append(i)
                        except StopIteration as fin:
                            done[i] = fin.value
                            alive.remove(i)
                if stacked and self._gate.ev is not None:
                    for i in stacked:
                        kids[i].stream.wait_event(self._gate.ev)
            rec = torch.empty((bsz, H, W, B), dtype=torch.float32, device=dev)
            norms = []
            for kid, ws, (lo, hi), (r, call, last, res_row) in zip(kids, wss, plan, done):
                with torch.cuda.stream(kid.stream):
                    ws.host_res.copy_(ws.res, non_blocking=True)
                    norms.append(ws.gram[:(hi - lo) * 80].view(hi - lo, 80)[:, 64:66].to("cpu", non_blocking=True))     # |F_k|^2, |G_k|^2 of the last solve (fp64)
                main.wait_stream(kid.stream)
                rec[lo:hi].copy_(r)
                r.record_stream(main)
            main.synchronize()
        except BaseException:
            for g in gens:
                g.close()
            torch.cuda.synchronize(dev)
            raise
        _, call, last, res_row = done[0]
        first = 2 if self.iterator == "anderson" else 1
        for ws in wss:
            if bool((ws.host_res[first:res_row, 0] < self.tol).any()):
                return None
        # the whole batch's residual of the last iteration, as K6's last block folds it: the samples' float64 norms added in sample order
        sf = sg = 0.0
        for nm in norms:
            for k in range(nm.shape[0]):
                sf += float(nm[k, 0])
                sg += float(nm[k, 1])
        eps = float(np.float32(1e-5 if self.iterator == "anderson" else 1e-7))     # (K6 takes eps as a float)
        res = float(np.float32(math.sqrt(sg) / (eps + math.sqrt(sf))))
        self._grouped = True
        self.den.stack_launches += max(k.den.stack_launches for k in kids)
        self.last_info = {"res": res, "res_per_sample": [v for ws in wss for v in ws.host_res[res_row, 1:].tolist()], "iterations": last,
                          "f_calls": call, "iterator": self.iterator, "graph": False, "groups": [list(p) for p in plan]}
        self._warn_if_not_finite()
        return rec

    def _warn_if_not_finite(self):
        """A non-finite residual is what an fp16 overflow inside the split-fp16 layers (activations beyond |x| = 255.9: csrc/conv_s16.hip keeps
        it inf / NaN all the way to the output on purpose) looks like from here - as does a genuinely diverging run.  Say so, loudly."""
        import warnings
        if not math.isfinite(self.last_info["res"]) and self.conv64_policy in ("fast", "s16"):
            what = ("redoing this call with conv64='fast32' (fp32 MFMA kernels, no such limit; last_info['conv64_fallback'])" if self.conv64 == "auto"
                    else "rerun with conv64='fast32' (fp32 MFMA kernels, no such limit)")
            how = ("16 x the largest activation of the first f-call, which set the scales" if self.den.act_range == "data" and self.den.ranges is not None
                   else "|x| >= 255.9 under act_range='fixed'; inputs are expected in [0, 1]")
            warnings.warn("DEQSCIEngine: the reconstruction's residual is not finite.  If the iteration itself is not diverging, an activation of "
                          f"the denoiser has left fp16's range inside the split-fp16 64->64 layers ({how}): " + what + ".", RuntimeWarning, stacklevel=4)

    def _enqueue(self, ws, y, Phi4, Phi_sum, initial_point, poll):
        """Every launch of one reconstruction on the current stream.  poll: True = lagged residual read-back with early stop,
        False = one read-back at the end, None = no host traffic at all (what a hipGraph capture records)."""
        steps = self._enqueue_steps(ws, y, Phi4, Phi_sum, initial_point, poll)
        while True:
            try:
                next(steps)
            except StopIteration as done:
                return done.value

    def _enqueue_steps(self, ws, y, Phi4, Phi_sum, initial_point, poll):
        """_enqueue as a generator that yields behind every f-call (a grouped reconstruction issues the f-calls of its halves alternately);
        returns what _enqueue returns."""
        self._eager = poll is not None
        phi = _hip.transpose(Phi4, LAYOUT_BHW)
        ps = Phi_sum if Phi_sum is not None else _hip.phi_sum(phi, LAYOUT_BHW)
        if initial_point is None:
            _hip.sci_adjoint(y, phi, LAYOUT_BHW, out=ws.xbuf[0])
        else:
            _hip.transpose(initial_point, LAYOUT_BHW, out=ws.xbuf[0])
        if self.iterator == "anderson":
            x_last, call, last = yield from self._anderson(ws, y, phi, ps, poll)
        else:
            x_last, call, last = yield from self._picard(ws, y, phi, ps, poll)
        res_row = last
        # z = f(z*)  (new_equilibrium_utils_yaping.py:268)
        _hip.gap_update(x_last, phi, y, ps, LAYOUT_BHW, out=ws.z1)
        if self.den.gate is not None:
            out, is_noise = yield from self.den.run_steps(ws.z1, call)
        else:
            out, is_noise = self.den.run(ws.z1, call)
        out = _hip.f32c(out)
        rec = _hip.residual_out(ws.z1, out, LAYOUT_HWB) if is_noise else _hip.transpose(out, LAYOUT_HWB)
        call += 1
        if self.extra_call:                                   # dead f0 = f(z) of :271-272
            yield "fcall"
            zt = _hip.transpose(rec, LAYOUT_BHW)
            _hip.gap_update(zt, phi, y, ps, LAYOUT_BHW, out=ws.z1)
            if self.den.gate is not None:
                yield from self.den.run_steps(ws.z1, call)
            else:
                self.den.run(ws.z1, call)
            call += 1
        return rec, call, last, res_row

    # ------------------------------------------------------------------ hipGraph replay of a whole reconstruction
    def _replay(self, ws, y, Phi4, Phi_sum, initial_point):
        """-> reconstruction, or None when this call has to take the eager path (first call of a shape: it warms every
        kernel up; or the tolerance test fired inside the replayed run)."""
        key = (tuple(y.shape), tuple(Phi4.shape), Phi_sum is not None, initial_point is not None, self.den._wkey, self.extra_call,
               None if self.den.sigma_table is None else self.den.sigma_table.data_ptr())
        g = self._graph
        if g is None or g["key"] != key or g["ws"] is not ws:
            # (the graph holds the workspace it was captured on: its nodes carry that workspace's raw pointers)
            self._graph = {"key": key, "ws": ws, "graph": None}
            return None                                       # eager now, capture on the next call with this key
        if g["graph"] is None:
            g["y"], g["Phi4"] = y.clone(), Phi4.clone()
            g["ps"] = None if Phi_sum is None else Phi_sum.clone()
            g["x0"] = None if initial_point is None else initial_point.clone()
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            c0 = self.den.stack_launches
            with torch.cuda.graph(graph):
                g["rec"], g["call"], g["last"], g["res_row"] = self._enqueue(ws, g["y"], g["Phi4"], g["ps"], g["x0"], None)
            g["graph"], g["stack_n"] = graph, self.den.stack_launches - c0
        else:
            self.den.stack_launches += g["stack_n"]
            g["y"].copy_(y)
            g["Phi4"].copy_(Phi4)
            if Phi_sum is not None:
                g["ps"].copy_(Phi_sum)
            if initial_point is not None:
                g["x0"].copy_(initial_point)
        g["graph"].replay()
        rec = g["rec"].clone()
        ws.host_res.copy_(ws.res, non_blocking=True)          # the whole residual table, once
        torch.cuda.current_stream().synchronize()
        first = 2 if self.iterator == "anderson" else 1
        table = ws.host_res[first:g["res_row"], 0]
        if bool((table < self.tol).any()):                    # the reference would have stopped earlier: redo it eagerly
            return None
        r = ws.host_res[g["res_row"]]
        self.last_info = {"res": float(r[0]), "res_per_sample": r[1:].tolist(), "iterations": g["last"],
                          "f_calls": g["call"], "iterator": self.iterator, "graph": True}
        self._warn_if_not_finite()
        return rec

    # ------------------------------------------------------------------ Anderson (new_equilibrium_utils_yaping.py:153-189)
    def _anderson(self, ws, y, phi, ps, poll):
        m, max_iter = self.m, self.max_iter
        if m < 2:
            raise IndexError("index 1 is out of bounds for dimension 1 with size %d" % m)   # X[:, 1] at :163
        xb = ws.xbuf
        # f-calls 1, 2 fill slots 0, 1 (:162-163)
        _hip.gap_update(xb[0], phi, y, ps, LAYOUT_BHW, out=ws.z1)
        yield from self._store_solve_steps(ws, xb[0], 0, 0, 1, 0, xb[1], 1e-5, 0)
        yield "fcall"
        _hip.gap_update(xb[1], phi, y, ps, LAYOUT_BHW, out=ws.z1)
        yield from self._store_solve_steps(ws, xb[1], 1, 1, 2, 2, None, 1e-5, 1)
        yield "fcall"
        if max_iter <= 2:
            raise UnboundLocalError("local variable 'res' referenced before assignment")      # :189 with the loop skipped
        last, prev_ev = None, None
        for k in range(2, max_iter):
            n = min(k, m)
            x = xb[k % 2]
            _hip.anderson_mix_gap(ws, self.beta, n, phi, y, ps, x, ws.z1, LAYOUT_BHW)
            nf = min(k + 1, m)
            yield from self._store_solve_steps(ws, x, k, k % m, nf, nf, None, 1e-5, k)
            last = k
            yield "fcall"
            if poll:
                ev = self._poll(ws, k)
                if prev_ev is not None:
                    prev_ev.synchronize()
                    if float(ws.host_res[k - 1, 0]) < self.tol:
                        last = k - 1
                        break
                prev_ev = ev
        if poll is not None and (not poll or last == max_iter - 1):
            ws.host_res[last].copy_(ws.res[last], non_blocking=True)
        return xb[last % 2], last + 1, last

    # ------------------------------------------------------------------ Picard (new_equilibrium_utils_yaping.py:213-222)
    def _picard(self, ws, y, phi, ps, poll):
        xb = ws.xbuf
        _hip.gap_update(xb[0], phi, y, ps, LAYOUT_BHW, out=ws.z1)
        yield from self._store_solve_steps(ws, xb[0], 0, 0, 1, 0, xb[1], 1e-7, 0)           # f0 = f(x0)
        yield "fcall"
        if self.max_iter <= 0:
            if poll is not None:
                ws.host_res[0].copy_(ws.res[0], non_blocking=True)
            return xb[1], 1, 0
        last, prev_ev = None, None
        for k in range(self.max_iter):
            x, nxt = xb[(k + 1) % 2], xb[k % 2]
            _hip.gap_update(x, phi, y, ps, LAYOUT_BHW, out=ws.z1)
            yield from self._store_solve_steps(ws, x, k + 1, 0, 1, 0, nxt, 1e-7, k + 1)
            last = k
            yield "fcall"
            if poll:
                ev = self._poll(ws, k + 1)
                if prev_ev is not None:
                    prev_ev.synchronize()
                    if float(ws.host_res[k, 0]) < self.tol:
                        last = k - 1
                        break
                prev_ev = ev
        if poll is not None and (not poll or last == self.max_iter - 1):
            ws.host_res[last + 1].copy_(ws.res[last + 1], non_blocking=True)
        return xb[last % 2], last + 2, last + 1

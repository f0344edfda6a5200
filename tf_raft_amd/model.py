"""Device mirror of reference ``tf_raft/model.py`` -- ``RAFT`` / ``SmallRAFT`` forward prediction.

Same constructor and call signature as the reference (model.py:11, 68, 174, 190):

    model = RAFT(drop_rate=0, iters=12, iters_pred=24)
    flow_predictions = model([image1, image2], training=False)   # list of (bs, H, W, 2)

``image1/2`` are ``(bs, H, W, 3)`` float arrays in 0..255 (NumPy or torch, host or device);
the result is a Python list of ``iters_pred`` (``iters`` when ``training=True``) fp32 device
tensors, ``[-1]`` the finest, each answering ``.numpy()`` like a TF eager tensor.

Execution: everything runs on hand-written HIP kernels behind the C ABI (``include/raft_hip.h``): the two
encoders (``raft_encoder_f32``), the correlation volume build, and the whole recurrent loop
(lookup -> update block -> coords update -> upsampling) enqueued by ONE ``raft_iterate_*`` call on
the current HIP stream with no host synchronisation.  There is no CPU fallback.

Callers of the forward pass (reference model.py:111-170): ``compile``, ``test_step`` (EPE / u1 / u3 / u5 of the final
prediction against ground truth, reduced on the device: ``tf_raft_amd.losses``), ``predict_step``, ``reset_metrics``,
``load_weights`` / ``save_weights`` (TensorFlow tensor-bundle checkpoints, read and written without TensorFlow) are
provided.  ``train_step`` (model.py:126-144) is functional for RAFT: training-mode forward, backward, global-norm clipping
and AdamW on HIP kernels (``tf_raft_amd.grad`` / ``tf_raft_amd.training``), orchestrated from Python -- not yet a tuned
path.
"""
from __future__ import annotations

import contextlib
import ctypes as C
import os
from collections import OrderedDict
from typing import Dict, Optional

import numpy as np
import torch

from . import _dev
from . import weights as weights_mod
from . import _ffi
from ._ffi import check
from .layers.corr import CorrBlock, coords_grid, upflow8
from .layers.extractor import BasicEncoder, SmallEncoder
from .layers.update import BasicUpdateBlock, SmallUpdateBlock, UpdateState



def _rank_of_this_process() -> int:
    """Data-parallel rank (0 without torch.distributed)."""
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


def _dropout_seed(model_seed: int, rank: int, step: int) -> int:
    """Even 62-bit seed of a training step's two dropout masks (the second uses seed + 1): a splitmix64-style mix of the
    model seed, the data-parallel rank and the step, so that ranks, models and steps draw independent masks."""
    x = (int(model_seed) * 0x9E3779B97F4A7C15 + int(rank) * 0xBF58476D1CE4E5B9 + int(step) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 30
    x = (x * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 27
    x = (x * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    x ^= x >> 31
    return int(x >> 2) & ~1


DEFAULT_LANES = 3          # profiles/r12b_lanes_ab_per_process.txt: 1 / 2 / 3 / 4 / 6 lanes = 332 / 352 / 362 / 347-352 / 353-362 pairs/s at 4 pairs


class RAFT:
    """reference model.py:10-109."""

    variant = 'raft'

    def __init__(self, drop_rate=0, iters=12, iters_pred=24, weights: Optional[Dict[str, np.ndarray]] = None,
                 seed=0, alternate_corr=False, overlap=None, pipeline=None, lanes=None, loop_concurrency=None, **kwargs):
        # reference model.py:11-12 forwards **kwargs to tf.keras.Model, whose constructor takes `name` (and nothing a
        # forward pass depends on): accept it, reject the rest
        self.name = kwargs.pop('name', type(self).__name__.lower())
        if kwargs:
            raise TypeError(f'unexpected keyword arguments {sorted(kwargs)}')
        self.hidden_dim = 128
        self.context_dim = 128
        self.corr_levels = 4
        self.corr_radius = 4
        self.drop_rate = drop_rate
        self.seed = int(seed)                       # also mixed into the dropout mask seeds of train_step
        self.iters = iters
        self.iters_pred = iters_pred
        self.alternate_corr = alternate_corr
        # three-stream schedule of the loop (RAFT only); RAFT_OVERLAP=0 forces the single-stream loop.  With several lanes
        # (below) the loops of a pipelined call default to the single-stream schedule: the other lanes fill the chain's idle CUs
        # and gaps better than a loop's own side branches do (profiles/r12b_lanes_ab_per_process.txt).
        env_ov = os.environ.get('RAFT_OVERLAP')
        self._overlap_given = overlap is not None or env_ov is not None
        self.overlap = (env_ov != '0') if overlap is None else bool(overlap)
        # Consecutive inference calls overlap (section "pipelined forward" below).  OPT-IN (pipeline=True or RAFT_PIPELINE=1):
        # a pipelined call returns before its loop has finished and its results join the consuming stream lazily, which
        # consumers that bypass torch's dispatch (C++ extensions taking at::Tensor, the legacy torch.utils.dlpack.to_dlpack)
        # cannot see -- the default is the serial schedule, where stream order alone makes every route to the bytes safe.
        # ``predict()`` pipelines regardless: it consumes its own results.
        self.pipeline = (os.environ.get('RAFT_PIPELINE', '0') == '1') if pipeline is None else bool(pipeline)
        # Pipelined forward: how many recurrent loops may be in flight at once (each on streams of its own; RAFT_LANES).
        self.lanes = max(1, int(os.environ.get('RAFT_LANES', str(DEFAULT_LANES))) if lanes is None else int(lanes))
        self._enc_stream = None
        self._state = None
        self._ring = []                          # pipelined forward: [UpdateState, loop-done event] per slot (lanes + 1 slots)
        self._calls = 0
        self._lane = 0                           # the lane whose streams / loop context the loop launchers use
        self._lane_mode = False                  # inside a pipelined call with several lanes and single-stream loops
        # which launches of a multi-lane call get the library's concurrency hint (launch shapes of a lanes-times larger batch):
        # 'loop' = the recurrent loop, 'all' = encoders and volume build as well, 'none'
        self._shape_hint = os.environ.get('RAFT_LANE_SHAPES', 'all')      # profiles/r12e_hint_ab.txt: 361 / 374 / 375 pairs/s with none / loop / all at 4 pairs
        # serial schedule: the hint its loops are launched with (1 = latency shapes; tests give a serial model the lanes of a
        # pipelined one to obtain the same kernels, hence the same bits)
        self.loop_concurrency = 1 if loop_concurrency is None else max(1, int(loop_concurrency))
        self._lane_res = {}                      # lane -> (device, (flow stream, mask stream), raft_loop_ctx handle)
        self._loop_stream = None
        _dev.reserve_streams(_dev.require_gpu())
        _dev.lib()
        if weights is None:
            weights = weights_mod.init_weights(self.variant, seed)     # Keras default initialisers
        weights_mod.check_weights(self.variant, weights)
        self._weights = dict(weights)
        self._dw, self._host_stale, self._train_vars = None, False, None     # device master copies of train_step
        self._build(weights)

    def _build(self, weights):
        self.fnet = BasicEncoder(output_dim=256, norm_type='instance', drop_rate=self.drop_rate,
                                 weights=weights, prefix='fnet')
        self.cnet = BasicEncoder(output_dim=self.hidden_dim + self.context_dim, norm_type='batch',
                                 drop_rate=self.drop_rate, weights=weights, prefix='cnet')
        self.update_block = BasicUpdateBlock(filters=self.hidden_dim, weights=weights, prefix='update_block')

    # ---- weights ------------------------------------------------------------------------------
    def set_weights(self, weights: Dict[str, np.ndarray]) -> None:
        weights_mod.check_weights(self.variant, weights)
        self._join_pipeline()                       # the old device blobs are freed below: no loop may still be reading them
        self._weights = dict(weights)
        self._inference_stale = False
        self._train_vars = None                     # train_step re-reads its device master copies
        self._dw, self._host_stale = None, False
        self.fnet.set_weights(weights)
        self.cnet.set_weights(weights)
        self.update_block.set_weights(weights)

    def load_weights(self, path: str) -> None:
        """reference README.md:66-96 / train_sintel.py:117-123: ``model.load_weights('checkpoints/model')``.
        ``path`` is a TensorFlow checkpoint prefix (``<path>.index`` + ``<path>.data-*``, read without TensorFlow by
        ``tf_raft_amd.checkpoint``) or a ``.npz`` written by ``save_weights`` (Keras layout either way)."""
        from . import checkpoint
        if checkpoint.is_tf_checkpoint(path):
            self.set_weights(checkpoint.load_tf_checkpoint(path, self.variant))
        else:
            self.set_weights(weights_mod.load_weights(path))

    def _sync_host(self):
        """``train_step`` keeps the master weights (and the batch-norm moving statistics) on the device and updates them in
        place; the NumPy dictionary is refreshed from them only when somebody asks for it (checkpoint, inference re-pack)."""
        if getattr(self, '_host_stale', False) and self._dw is not None:
            self._weights = {k: v.detach().cpu().numpy() for k, v in self._dw.items()}
            self._host_stale = False

    def get_weights_dict(self) -> Dict[str, np.ndarray]:
        """The current weights in Keras layout under ``tf_raft_amd.weights`` names."""
        self._sync_host()
        return dict(self._weights)

    def save_weights(self, path: str) -> None:
        """reference train_sintel.py:104-107 (ModelCheckpoint(save_weights_only=True)): ``path`` ending in ``.npz``
        writes the NumPy container, anything else a TensorFlow tensor-bundle checkpoint prefix."""
        from . import checkpoint
        self._sync_host()
        if path.endswith('.npz'):
            weights_mod.save_weights(path, self._weights)
        else:
            checkpoint.write_tf_checkpoint(path, self._weights, self.variant)

    # ---- reference helpers ----------------------------------------------------------------------
    def initialize_flow(self, image):
        """reference model.py:32-37."""
        bs, h, w, _ = image.shape
        return coords_grid(bs, h // 8, w // 8), coords_grid(bs, h // 8, w // 8)

    def upsample_flow(self, flow, mask):
        """reference model.py:39-66.  flow (bs, h, w, 2), mask (bs, h, w, 576) -> (bs, 8h, 8w, 2)."""
        flow = _dev.to_device(flow)
        mask = _dev.to_device(mask)
        bs, h, w, _ = flow.shape
        if tuple(mask.shape) != (bs, h, w, 576) or flow.shape[-1] != 2:
            raise ValueError(f'expected flow (bs,h,w,2) and mask (bs,h,w,576), got {tuple(flow.shape)}, {tuple(mask.shape)}')
        out = torch.empty((bs, 8 * h, 8 * w, 2), device=flow.device, dtype=torch.float32)
        check(_dev.lib().raft_upsample_convex_f32(_dev.ptr(flow), _dev.ptr(mask), bs, h, w, _dev.ptr(out),
                                                  _dev.stream_ptr()), 'upsample_convex')
        return _dev.wrap(out)

    # ---- forward ----------------------------------------------------------------------------------
    def _get_state(self, B, h, w, device) -> UpdateState:
        st = self._state
        if st is None or (st.B, st.h, st.w) != (B, h, w) or st.net.device != device:
            st = UpdateState(self.variant, B, h, w, device)
            self._state = st
        return st

    def _prepare(self, cnet, st):
        check(_dev.lib().raft_prepare_state_f32(_dev.ptr(cnet), st.B, st.h, st.w, C.byref(st.c),
                                                _dev.stream_ptr()), 'prepare_state')
        self.update_block.prepare(st)          # GRU terms of `inp`: constant over the loop (model.py:86)

    @staticmethod
    def _lane_role(role, lane):
        return role if lane == 0 else f'{role}{lane}'

    def _loop_priority(self):
        """Stream priority of the loop lanes.  RAFT_LOOP_PRIORITY=1 (high) is worth +0.7 % in a process that creates nothing but one
        multi-lane model (379.4 / 380.0 against 376.9 / 377.0 pairs/s, A/B/A/B, profiles/r12n_steps_and_priority.txt) and costs 10 - 70 %
        in a process that also holds a serial model: priority streams come from a pool of their own, and the two pools' streams then
        collide on HIP's four hardware queues (profiles/r12t_config_bench.txt: SmallRAFT 747 -> 597, a serial model after a multi-lane one
        328 -> 187 pairs/s).  Not the default."""
        return -1 if os.environ.get('RAFT_LOOP_PRIORITY', '0') == '1' else _dev.STREAM_PRIORITY

    def _lane_entry(self, dev):
        """(device, (flow stream, mask stream), loop-context handle or None) of the current lane."""
        ent = self._lane_res.get(self._lane)
        if ent is None or ent[0] != dev:
            if ent is not None:
                self._free_lane(self._lane)
            prio = self._loop_priority()
            aux = (_dev.side_stream(dev, self._lane_role('flow', self._lane), prio),
                   _dev.side_stream(dev, self._lane_role('mask', self._lane), prio))
            ent = self._lane_res[self._lane] = [dev, aux, None]
        return ent

    def _aux_streams(self, dev):
        return self._lane_entry(dev)[1]

    @contextlib.contextmanager
    def _capturable_stream(self, dev):
        """The three-stream loops replay a captured hipGraph for small batches (include/raft_hip.h, RAFT_LOOP_GRAPH); the
        legacy default stream (handle 0, torch's default) cannot be captured, so when the caller is on it the loop is
        enqueued on a private stream ordered after / before the current one."""
        cur = torch.cuda.current_stream(dev)
        if cur.cuda_stream != 0 or _ffi.get_option('RAFT_LOOP_GRAPH') != '1':    # replay is opt-in (measured slower)
            yield
            return
        if self._loop_stream is None or self._loop_stream.device != dev:
            self._loop_stream = _dev.side_stream(dev, 'loop')
        self._loop_stream.wait_stream(cur)
        with torch.cuda.stream(self._loop_stream):
            yield
        cur.wait_stream(self._loop_stream)   # joined: buffers allocated on `cur` are safe to reuse after this point

    def _loop_context(self, dev):
        """The caller-owned ``raft_loop_ctx`` (cross-stream events + hipGraph cache) of the three-stream loops: one per lane
        (loops of different lanes run concurrently; loops of one lane follow each other in stream order and share it)."""
        ent = self._lane_entry(dev)
        if ent[2] is None:
            handle = C.c_void_p()
            with torch.cuda.device(dev):
                check(_dev.lib().raft_loop_ctx_create(C.byref(handle)), 'loop_ctx_create')
            ent[2] = handle
        return ent[2]

    def _free_lane(self, lane):
        ent = self._lane_res.pop(lane, None)
        if ent is not None and ent[2] is not None:
            try:
                torch.cuda.synchronize(ent[0])
                _dev.lib().raft_loop_ctx_destroy(ent[2])
            except Exception:   # noqa: BLE001  (interpreter shutdown)
                pass

    def _free_loop_context(self):
        for lane in list(getattr(self, '_lane_res', {})):
            self._free_lane(lane)

    def __del__(self):
        self._free_loop_context()

    def _iterate(self, corr: CorrBlock, st, iters, flow_up):
        if self.overlap:
            # flow branch and mask branch of every iteration on two side streams (events inside the library)
            dev = flow_up.device
            aux = self._aux_streams(dev)
            with self._capturable_stream(dev):
                check(_dev.lib().raft_iterate_basic_overlap_f32(
                    C.byref(self.update_block.c), _dev.ptr(corr._pyr), corr._off, st.B, st.h, st.w, iters, C.byref(st.c),
                    _dev.ptr(flow_up), _dev.stream_ptr(), aux[0].cuda_stream, aux[1].cuda_stream,
                    self._loop_context(dev)), 'iterate_basic_overlap')
            return
        check(_dev.lib().raft_iterate_basic_f32(C.byref(self.update_block.c), _dev.ptr(corr._pyr), corr._off,
                                                st.B, st.h, st.w, iters, C.byref(st.c), _dev.ptr(flow_up),
                                                _dev.stream_ptr()), 'iterate_basic')

    def _iterate_alternate(self, corr: CorrBlock, st, iters, flow_up):
        if self.variant == 'raft' and (self.overlap or self._lane_mode):
            # the same C loop (three streams, or the loop's own stream three times = the single-stream schedule of a lane),
            # lookups computed on demand from fmap1 and the pooled fmap2 pyramid
            dev = flow_up.device
            with self._capturable_stream(dev):
                s0 = _dev.stream_ptr()
                s1, s2 = ((a.cuda_stream for a in self._aux_streams(dev)) if self.overlap else (s0, s0))
                check(_dev.lib().raft_iterate_basic_ondemand_f32(
                    C.byref(self.update_block.c), _dev.ptr(corr.fmap1), _dev.ptr(corr._f2pyr), corr.fmap1.shape[-1],
                    st.B, st.h, st.w, iters, C.byref(st.c), _dev.ptr(flow_up), s0, s1, s2, self._loop_context(dev)),
                    'iterate_basic_ondemand')
            return
        g = st.g
        for i in range(iters):
            corr.retrieve(st.coords1, out=st.corr, ld_out=g['corr_ld'])
            self.update_block.step(st)
            self._upsample_into(st, flow_up[i])

    def _upsample_into(self, st, out):
        check(_dev.lib().raft_upsample_convex_f32(_dev.ptr(st.flow), _dev.ptr(st.mask), st.B, st.h, st.w,
                                                  _dev.ptr(out), _dev.stream_ptr()), 'upsample_convex')

    def __call__(self, inputs, training=False):
        return self.call(inputs, training)

    def call(self, inputs, training=False):
        """reference model.py:68-109."""
        return self._forward(inputs, training)

    def _sync_inference_weights(self):
        if getattr(self, '_inference_stale', False):
            self._sync_host()
            keep = (self._train_vars, self._dw)
            self.set_weights(self._weights)
            self._train_vars, self._dw = keep
            self._inference_stale = False

    def _forward(self, inputs, training=False, final_only=False, pipelined=None):
        self._sync_inference_weights()
        image1, image2 = inputs
        image1 = _dev.to_device(image1)
        image2 = _dev.to_device(image2)
        if image1.dim() != 4 or image1.shape[-1] != 3 or image1.shape != image2.shape:
            raise ValueError(f'images must both be (bs, H, W, 3), got {tuple(image1.shape)} / {tuple(image2.shape)}')
        B, H, W, _ = image1.shape
        if H % 8 or W % 8:
            raise ValueError(f'H and W must be multiples of 8 (got {H}x{W})')   # model.py:35 uses h//8
        # model.py:70-71 (2 * (image / 255) - 1) is applied by the encoders while they stage the image
        if (self.pipeline if pipelined is None else pipelined) and not training:
            # several lanes (and their launch shapes) only for a model that asked for the pipelined schedule: predict() on a serial
            # model overlaps its calls on ONE lane with the serial schedule's kernels, so predict() == predict_step() bit for bit
            return self._forward_pipelined(image1, image2, final_only, self.lanes if self.pipeline else 1)
        self._join_pipeline()                       # (a training-mode or serial call after pipelined ones)
        self._lane = 0
        if self.loop_concurrency > 1 and self._shape_hint == 'all' and not training:
            with _ffi.thread_concurrency(self.loop_concurrency):        # the whole call with a multi-lane call's launch shapes
                return self._forward_serial(image1, image2, training, final_only)
        return self._forward_serial(image1, image2, training, final_only)

    def _forward_serial(self, image1, image2, training, final_only):
        B, H, W, _ = image1.shape
        if self.overlap and not training:
            # the context encoder does not depend on the feature encoder or the volume: it runs on a side stream
            # next to them (its one-workgroup-per-CU layers fill the tails of the feature encoder's launches)
            cur = torch.cuda.current_stream(image1.device)
            if self._enc_stream is None or self._enc_stream.device != image1.device:
                self._enc_stream = _dev.side_stream(image1.device, 'encoder')
            st = self._get_state(B, H // 8, W // 8, image1.device)    # (allocated under the caller's stream, like its other users)
            self._enc_stream.wait_stream(cur)      # also orders this call's state preparation behind the previous call's loop
            with torch.cuda.stream(self._enc_stream):
                cnet = self.cnet(image1, training=training, _raw_images=True)      # model.py:82
                # net / inp / the GRU's context rows depend on cnet only (model.py:84-89): prepared here, beside the feature
                # encoder, instead of behind the volume build (0.1 ms of the step at 4 pairs)
                self._prepare(cnet, st)
        fmap1, fmap2 = self.fnet([image1, image2], training=training, _raw_images=True)   # model.py:74
        correlation = CorrBlock(fmap1, fmap2, num_levels=self.corr_levels, radius=self.corr_radius,
                                alternate=self.alternate_corr)                  # model.py:77
        h, w = H // 8, W // 8
        if self.overlap and not training:
            cur.wait_stream(self._enc_stream)
            cnet.as_subclass(torch.Tensor).record_stream(cur)
        else:
            cnet = self.cnet(image1, training=training, _raw_images=True)      # model.py:82
            st = self._get_state(B, h, w, image1.device)
            self._prepare(cnet, st)                                             # model.py:84-89
        iters = self.iters if training else self.iters_pred
        with _ffi.thread_concurrency(self.loop_concurrency if (not training and self._shape_hint in ('loop', 'all')) else 1):
            out = self._run_loop(correlation, st, iters, self._alloc_out(iters, B, H, W, image1.device, final_only), final_only)
        return _dev.wrap(out) if final_only else [_dev.wrap(out[i]) for i in range(iters)]   # model.py:109

    @staticmethod
    def _alloc_out(iters, B, H, W, dev, final_only):
        return torch.empty((B, H, W, 2) if final_only else (iters, B, H, W, 2), device=dev, dtype=torch.float32)

    def _run_loop(self, correlation, st, iters, out, final_only):
        """model.py:93-109 on the CURRENT stream (+ the two aux streams of the three-stream schedule) into ``out``."""
        B, h, w = st.B, st.h, st.w
        if final_only:
            last = out
            with self._capturable_stream(last.device):
                # single-stream schedule (several lanes): the flow / mask "branches" are the loop's own stream
                s0 = _dev.stream_ptr()
                s1, s2 = ((a.cuda_stream for a in self._aux_streams(last.device)) if self.overlap else (s0, s0))
                check(_dev.lib().raft_iterate_basic_final_f32(
                    C.byref(self.update_block.c), _dev.ptr(correlation._pyr), correlation._off, B, h, w, iters, C.byref(st.c),
                    _dev.ptr(last), s0, s1, s2, self._loop_context(last.device)), 'iterate_basic_final')
            self._last_correlation = correlation
            return last
        flow_up = out
        if self.alternate_corr:
            self._iterate_alternate(correlation, st, iters, flow_up)
        else:
            self._iterate(correlation, st, iters, flow_up)                      # model.py:93-106
        self._last_correlation = correlation                                    # keep buffers alive until the stream drains
        return flow_up

    # ---- pipelined forward ------------------------------------------------------------------------------------------------
    # One inference call = encoders + volume build (2.96 of 12.3 ms at 4 pairs, kernels that fill the chip) followed by the 24
    # iterations of a DEPENDENT chain whose launches leave 32 .. 88 of the 256 CUs idle.  Back-to-back calls are independent of
    # each other, so call n + 1's pre-loop work can run under call n's loop: the loop is enqueued on the process-wide 'loop'
    # stream (its flow / mask branches on the aux streams, as before), the pre-loop work stays on the caller's stream, and the
    # caller's stream is NOT made to wait for the loop -- the returned tensors carry a `_dev.Pending` and whichever stream first
    # touches their data waits for the loop then (a caller that consumes the result at once sees the serial schedule).  What
    # makes this safe: the loop's inputs that the next call would overwrite exist twice (UpdateState ring; call n + 2's
    # pre-loop waits for loop n's event before it touches slot n & 1), per-call allocations read or written by the loop
    # (volume, feature maps of the on-demand lookup, the predictions) are recorded on the loop stream so the caching allocator
    # cannot hand them out before the loop has finished, and loops of consecutive calls follow each other in stream order.
    # Per-call results are bit-identical to the serial schedule (same kernels, same order per call):
    # tests/test_gpu_model.py::test_pipelined_calls_are_bitwise_the_serial_calls.
    def _join_pipeline(self):
        """Make the current stream wait for every loop this model still has in flight: called before anything that frees or
        rewrites buffers a loop reads (weight blobs, training's in-place optimizer updates) -- in the serial schedule stream order
        gave that for free."""
        for ent in getattr(self, '_ring', ()):
            if ent is not None and ent[1] is not None:
                torch.cuda.current_stream(ent[0].net.device).wait_event(ent[1])

    def _ring_state(self, slot, B, h, w, device):
        while len(self._ring) <= slot:
            self._ring.append(None)
        ent = self._ring[slot]
        if ent is None or (ent[0].B, ent[0].h, ent[0].w) != (B, h, w) or ent[0].net.device != device:
            if ent is not None and ent[1] is not None:
                ent[1].synchronize()                     # the old buffers are about to be freed: their last loop must be done
            ent = self._ring[slot] = [UpdateState(self.variant, B, h, w, device), None]
        return ent

    # Several loops in flight (round 6).  The loop's kernels are launched for ONE batch: 7 * 2^k workgroups on 256 CUs, a
    # ~3 us boundary between dependent kernels, one event gap per iteration -- which is why 8 pairs per call run 9 % and 16 pairs
    # 17 % faster per pair than 4.  With `lanes` = D > 1 call n's loop runs on lane n % D (loop / flow / mask streams and a
    # raft_loop_ctx of its own), so up to D loops of consecutive calls are resident together and each fills the other's gaps and
    # idle CUs; the UpdateState ring has D + 1 slots (call n + D + 1's pre-loop waits for loop n).  Each call still runs exactly
    # the kernels of the serial schedule in the same order on its own buffers: results stay bit-identical per call.
    def _forward_pipelined(self, image1, image2, final_only, lanes):
        B, H, W, _ = image1.shape
        h, w = H // 8, W // 8
        dev = image1.device
        cur = torch.cuda.current_stream(dev)
        n = self._calls
        self._calls += 1
        self._lane = lane = n % lanes
        keep_overlap = self.overlap
        if lanes > 1 and not self._overlap_given:
            self.overlap = False                         # single-stream loops: the lanes are each other's side branches
            self._lane_mode = True
        try:
            if lanes > 1 and self._shape_hint == 'all':
                with _ffi.thread_concurrency(self._hint_value(lanes)):
                    return self._forward_lane(image1, image2, final_only, n, lane, lanes, cur, dev, B, H, W, h, w)
            return self._forward_lane(image1, image2, final_only, n, lane, lanes, cur, dev, B, H, W, h, w)
        finally:
            self.overlap = keep_overlap
            self._lane = 0
            self._lane_mode = False

    @staticmethod
    def _hint_value(lanes):
        """The launch-shape hint of a multi-lane call: the number of loops that share the chip (RAFT_LANE_HINT overrides, for A/B runs)."""
        v = os.environ.get('RAFT_LANE_HINT')
        return max(1, int(v)) if v else lanes

    def _forward_lane(self, image1, image2, final_only, n, lane, lanes, cur, dev, B, H, W, h, w):
        loop = _dev.side_stream(dev, self._lane_role('loop', lane), priority=self._loop_priority())
        ent = self._ring_state(n % (lanes + 1), B, h, w, dev)
        st = ent[0]
        if ent[1] is not None:
            cur.wait_event(ent[1])                       # slot's previous user (call n - lanes - 1): its loop read this state
        cnet = self.cnet(image1, training=False, _raw_images=True)                      # model.py:82
        self._prepare(cnet, st)                                                          # model.py:84-89
        fmap1, fmap2 = self.fnet([image1, image2], training=False, _raw_images=True)    # model.py:74
        correlation = CorrBlock(fmap1, fmap2, num_levels=self.corr_levels, radius=self.corr_radius,
                                alternate=self.alternate_corr)                           # model.py:77
        out = self._alloc_out(self.iters_pred, B, H, W, dev, final_only)      # from the caller's stream's pool, like every other buffer
        ready = torch.cuda.Event()
        ready.record(cur)
        loop.wait_event(ready)
        with torch.cuda.stream(loop), _ffi.thread_concurrency(self._hint_value(lanes) if self._shape_hint in ('loop', 'all') else 1):
            self._run_loop(correlation, st, self.iters_pred, out, final_only)
            done = torch.cuda.Event()
            done.record(loop)
        for t in (out, getattr(correlation, '_pyr', None), getattr(correlation, '_f2pyr', None), correlation.fmap1, correlation.fmap2):
            if isinstance(t, torch.Tensor) and t.is_cuda:
                t.as_subclass(torch.Tensor).record_stream(loop)      # allocated under `cur`, in use on `loop`
        ent[1] = done
        pending = _dev.Pending(done, dev)
        if final_only:
            return _dev.wrap(out, pending)
        return [_dev.wrap(out[i], pending) for i in range(self.iters_pred)]             # model.py:109

    def predict_step(self, data, _pipelined=None):
        """reference model.py:160-166: ``flow_predictions[-1]`` of the forward pass.  RAFT computes it with the mask head
        and the convex upsampling in the last iteration only (``raft_iterate_basic_final_f32``: the recurrence itself
        is unchanged, so the result equals ``self(...)[-1]``)."""
        image1, image2, *_ = data
        if self.variant == 'raft' and self.overlap and not self.alternate_corr:
            return self._forward([image1, image2], training=False, final_only=True, pipelined=_pipelined)
        return self._forward([image1, image2], training=False, pipelined=_pipelined)[-1]

    def predict(self, x, batch_size=None, steps=None, **kwargs):
        """``keras.Model.predict`` over ``predict_step`` (reference model.py:160-166): the final flow of every image
        pair as one host array ``(N, H, W, 2)``.

        ``x`` is ``[image1, image2]`` (host or device arrays ``(N, H, W, 3)``, split into batches of ``batch_size``,
        Keras' default 32) or an iterable of ``(image1, image2, ...)`` batches (the reference's ``tf.data`` datasets).
        Host batches are uploaded one batch ahead of the compute stream and the predictions come back through pinned
        buffers one batch behind it (``tf_raft_amd.prefetch``), so neither transfer sits on the critical path."""
        from .prefetch import prefetch_to_device
        dev = _dev.require_gpu()
        arrays = isinstance(x, (list, tuple)) and len(x) == 2 and all(getattr(a, 'ndim', 0) == 4 for a in x)
        if arrays:
            n = x[0].shape[0]
            if x[1].shape[0] != n:
                raise ValueError(f'image1 and image2 hold {n} and {x[1].shape[0]} images')
            bs = int(batch_size) if batch_size else 32
            if bs < 1:
                raise ValueError(f'batch_size must be >= 1, got {batch_size}')
            batches = ((x[0][i:i + bs], x[1][i:i + bs]) for i in range(0, n, bs))
        else:
            batches = iter(x)
        if steps is not None:
            import itertools
            batches = itertools.islice(batches, int(steps))
        batches = ((b[0], b[1]) for b in batches)
        down = torch.cuda.Stream(device=dev)
        pins, landing, results = {}, [], []
        total = n if (arrays and steps is None) else None
        whole, filled = None, 0                  # one preallocated host array when the number of pairs is known

        def collect():
            nonlocal whole, filled
            pin, ev = landing.pop(0)
            ev.synchronize()
            got = pin.numpy()
            if total is None:
                results.append(got.copy())
                return
            if whole is None:
                whole = np.empty((total,) + got.shape[1:], got.dtype)
            whole[filled:filled + got.shape[0]] = got
            filled += got.shape[0]

        for i, (image1, image2) in enumerate(prefetch_to_device(batches, buffer_size=1, device=dev)):
            res = self.predict_step((image1, image2), _pipelined=True)   # consumed below on `down`: the compute stream never waits for a loop
            cur = torch.cuda.current_stream(dev)
            key = (i & 1, tuple(res.shape))
            if key not in pins:
                pins[key] = torch.empty(res.shape, dtype=res.dtype).pin_memory()
            down.wait_stream(cur)
            with torch.cuda.stream(down):
                out = res.as_subclass(torch.Tensor)      # pipelined forward: `down`, not the compute stream, waits for the loop
                pins[key].copy_(out, non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(down)
            out.record_stream(down)
            landing.append((pins[key], ev))
            if len(landing) > 1:
                collect()
        while landing:
            collect()
        if whole is not None:
            return whole
        if not results:
            raise ValueError('predict() received no batches')
        return np.concatenate(results, axis=0)

    # ---- evaluation plumbing (reference model.py:111-170)
    def compile(self, optimizer=None, clip_norm=None, loss=None, epe=None, trainable='all', tape_dtype='f32', **kwargs):
        """reference model.py:111-124.  ``loss`` / ``epe`` default to ``tf_raft_amd.losses.sequence_loss`` /
        ``end_point_error``.  ``trainable``: which weights ``train_step`` updates -- ``'all'`` (the reference's behaviour:
        encoders, norms and update block) or ``'update_block'`` (encoders frozen, run by the inference kernels)."""
        from . import losses
        if kwargs:
            raise TypeError(f'unexpected keyword arguments {sorted(kwargs)}')
        if trainable not in ('all', 'update_block'):
            raise ValueError(f"trainable must be 'all' or 'update_block', got {trainable!r}")
        if tape_dtype not in ('f32', 'bf16'):     # bf16: the loop's activation tape is STORED as bf16 (BASELINE configs[4]), fp32 arithmetic
            raise ValueError(f"tape_dtype must be 'f32' or 'bf16', got {tape_dtype!r}")
        self.optimizer = optimizer
        self.clip_norm = clip_norm
        self.loss = loss if loss is not None else losses.sequence_loss
        self.epe = epe if epe is not None else losses.end_point_error
        self.trainable = trainable
        self._train_vars = None
        self.tape_dtype = tape_dtype
        self.flow_metrics = OrderedDict((k, losses.Mean(name=k)) for k in ('loss', 'epe', 'u1', 'u3', 'u5'))

    BN_MOMENTUM = 0.99      # Keras BatchNormalization default (extractor.py:10 passes none)

    def train_step(self, data):
        """reference model.py:126-144: forward with ``iters`` iterations in training mode, ``sequence_loss``, backward,
        ``clip_by_global_norm``, the optimizer's ``apply_gradients``, metrics.  Everything arithmetic is a HIP kernel
        (``tf_raft_amd.grad``: encoders in training form with instance / batch-statistics norms, volume build and its
        backward, the loop and its backward through time; ``tf_raft_amd.training.AdamW``).  With
        ``compile(..., trainable='update_block')`` the encoders stay frozen and run on the inference kernels.
        The step is orchestrated from Python; the updated master weights are re-packed for the kernels ON THE DEVICE, one launch
        per layer and use (``raft_pack_train_conv_f32``).  It is the functional path (parity-tested against autograd on the
        oracle), not yet a tuned one.  RAFT and SmallRAFT."""
        from . import grad, losses
        if not hasattr(self, 'flow_metrics'):
            raise RuntimeError('call compile() before train_step()')
        if self.loss is not losses.sequence_loss:
            raise NotImplementedError('train_step differentiates tf_raft_amd.losses.sequence_loss only')
        self._join_pipeline()                       # inference loops still in flight read the weights this step updates in place
        if self.optimizer is None or not hasattr(self.optimizer, 'apply_gradients'):
            raise RuntimeError('compile() needs an optimizer with apply_gradients(grads, variables, clip_norm) '
                               '(tf_raft_amd.training.AdamW)')
        image1, image2, flow, valid = data
        image1 = _dev.to_device(image1).as_subclass(torch.Tensor).to(torch.float32)
        image2 = _dev.to_device(image2).as_subclass(torch.Tensor).to(torch.float32)
        # the ground truth goes to the device NOW: an upload from pageable host memory waits for everything already queued on
        # the stream, and in front of the loss that is the whole forward pass (the host would enqueue the backward only after
        # the GPU had drained: 20 ms of a 85 ms step)
        flow, valid = losses._truth((flow, valid))
        B, H, W, _ = image1.shape
        if H % 8 or W % 8:
            raise ValueError(f'H and W must be multiples of 8 (got {H}x{W})')
        h, w = H // 8, W // 8
        full = self.trainable == 'all'
        # Master copies of EVERY weight (and the batch-norm moving statistics) live on the device from the first step on and
        # are updated in place; grad.* packs them for the convolution kernels on the device too.  Nothing of a step crosses
        # PCIe except the input batch (get_weights_dict / save_weights / the next inference call pull them back lazily).
        if self._dw is None:
            self._dw = {k: _dev.to_device(np.ascontiguousarray(v, dtype=np.float32)).as_subclass(torch.Tensor).clone()
                        for k, v in self._weights.items()}
        wts = self._dw
        grad.clear_pack_cache()                     # packed copies of last step's weights
        drop_masks = []
        if full:
            ones = torch.ones_like(image1)
            x1 = grad._axpby(2.0 / 255.0, image1.contiguous(), -1.0, ones)            # model.py:70-71
            x2 = grad._axpby(2.0 / 255.0, image2.contiguous(), -1.0, ones)
            fout, ftape = grad.encoder_forward(wts, 'fnet', torch.cat([x1, x2], dim=0), training=True)     # model.py:74
            fout = fout.as_subclass(torch.Tensor)
            cnet, ctape = grad.encoder_forward(wts, 'cnet', x1, training=True)         # model.py:82
            cnet = cnet.as_subclass(torch.Tensor)
            if self.drop_rate:     # extractor.py:109-111, 127-128: Dropout on the encoder outputs while training
                # mask seeds: a different stream per data-parallel rank (ranks see different shards), per model seed and per
                # optimizer step.  save_weights / load_weights persist the WEIGHTS only (as the reference's
                # ModelCheckpoint(save_weights_only=True) does, train_sintel.py:104-107): a caller that resumes a run must restore
                # optimizer.iterations itself, otherwise the mask sequence (and the LR schedule) restarts at step 1
                step = int(getattr(getattr(self, 'optimizer', None), 'iterations', 0) or 0)
                self._drop_step = max(getattr(self, '_drop_step', 0) + 1, step + 1)
                base = _dropout_seed(getattr(self, 'seed', 0), _rank_of_this_process(), self._drop_step)
                fout, mf = grad.dropout_forward(fout, self.drop_rate, seed=base)
                cnet, mc = grad.dropout_forward(cnet, self.drop_rate, seed=base + 1)
                drop_masks = [mf, mc]
            fmap1, fmap2 = fout[:B].contiguous(), fout[B:].contiguous()
        else:
            if self.drop_rate:
                raise NotImplementedError("drop_rate > 0 with trainable='update_block': dropout sits on the (frozen) encoder "
                                          "outputs; train all weights or use drop_rate=0")
            fmap1, fmap2 = self.fnet([image1, image2], training=False, _raw_images=True)
            cnet = self.cnet(image1, training=False, _raw_images=True)
        correlation = CorrBlock(fmap1, fmap2, num_levels=self.corr_levels, radius=self.corr_radius)   # model.py:77
        st = self._get_state(B, h, w, image1.device)
        prep = _dev.lib().raft_prepare_state_f32 if self.variant == 'raft' else _dev.lib().raft_prepare_state_small_f32
        check(prep(_dev.ptr(cnet), st.B, st.h, st.w, C.byref(st.c), _dev.stream_ptr()), 'prepare_state')   # model.py:84-86 / 209-211
        net0 = st.net.clone()
        inp = st.x[..., :self.context_dim].contiguous()
        prefix = 'update_block'
        ub = {k: v for k, v in wts.items() if k.startswith(prefix)}
        preds, tape = grad.loop_forward(ub, correlation, net0, inp, self.iters, prefix, self.variant, tape_dtype=self.tape_dtype)
        loss = self.loss([flow, valid], preds)
        d_preds = grad.sequence_loss_grad((flow, valid), preds)
        d_net0, d_inp, d_pyr, grads = grad.loop_backward(ub, correlation, tape, d_preds, prefix)
        stats = {}
        if full:
            d_f1, d_f2 = grad.corr_build_backward(correlation, d_pyr)
            d_fout = torch.cat([d_f1.as_subclass(torch.Tensor), d_f2.as_subclass(torch.Tensor)], dim=0).contiguous()
            d_cnet = grad.prepare_state_backward(net0, inp, d_net0, d_inp)
            if drop_masks:
                d_fout = grad.dropout_backward(d_fout, drop_masks[0])
                d_cnet = grad.dropout_backward(d_cnet.as_subclass(torch.Tensor).contiguous(), drop_masks[1])
            gf, _ = grad.encoder_backward(wts, 'fnet', ftape, d_fout)
            gc, stats = grad.encoder_backward(wts, 'cnet', ctape, d_cnet)
            grads = dict(grads)
            grads.update(gf)
            grads.update(gc)
        names = sorted(grads)
        self._train_vars = {k: wts[k] for k in names}          # views of the device master copies (updated in place)
        gt = {k: grads[k].as_subclass(torch.Tensor).reshape(wts[k].shape).contiguous() for k in names}
        from .parallel import all_reduce_gradients, all_reduce_mean_
        all_reduce_gradients(gt)                    # data-parallel training: one bucketed RCCL all-reduce (no-op on one rank)
        self.optimizer.apply_gradients(gt, self._train_vars, clip_norm=self.clip_norm)
        # Keras BatchNormalization moving statistics (momentum 0.99), in place on the device.  TF 2.3's fused kernel feeds the
        # moving variance with the UNBIASED batch variance; that detail cannot be checked here (no TensorFlow) and is stated in
        # DESIGN.md.  Data-parallel: the batch statistics are averaged over the ranks first (one small all-reduce), so that
        # every rank keeps the same moving statistics and a checkpoint does not depend on which rank writes it.
        if stats:
            order = sorted(stats)
            all_reduce_mean_([t for name in order for t in (stats[name][0], stats[name][1])])
            for name in order:
                mean, var, cnt = stats[name]
                mm, mv = wts[f'{name}/moving_mean'], wts[f'{name}/moving_variance']
                mm.copy_(grad._axpby(self.BN_MOMENTUM, mm, 1.0 - self.BN_MOMENTUM, mean.contiguous()))
                mv.copy_(grad._axpby(self.BN_MOMENTUM, mv, (1.0 - self.BN_MOMENTUM) * cnt / max(cnt - 1.0, 1.0), var.contiguous()))
            grad.bump_pack_version()                     # parameters written outside the optimizer: packed copies are stale
        # the NumPy dictionary and the inference kernels' re-packed (Winograd-transformed, N-fused) copies are refreshed
        # lazily: by get_weights_dict / save_weights, and by the next forward call
        self._host_stale = True
        self._inference_stale = True
        info = self.epe([flow, valid], preds[-1])
        self.flow_metrics['loss'].update_state(loss)
        for k in ('epe', 'u1', 'u3', 'u5'):
            self.flow_metrics[k].update_state(info[k])
        return {k: m.result() for k, m in self.flow_metrics.items()}

    def test_step(self, data):
        """reference model.py:146-158: forward prediction, then EPE / u1 / u3 / u5 of ``flow_predictions[-1]`` against
        ``(flow, valid)`` into the running means; returns ``{name: mean so far}`` (``loss`` is only fed by train_step).
        Only the last prediction is consumed, so it is computed the ``predict_step`` way."""
        if not hasattr(self, 'flow_metrics'):
            raise RuntimeError('call compile() before test_step()')   # keras raises for an un-compiled model too
        image1, image2, flow, valid = data
        last = self.predict_step((image1, image2))
        info = self.epe([flow, valid], last)
        for k in ('epe', 'u1', 'u3', 'u5'):
            self.flow_metrics[k].update_state(info[k])
        return {k: m.result() for k, m in self.flow_metrics.items()}

    def reset_metrics(self):
        """reference model.py:168-170."""
        for m in getattr(self, 'flow_metrics', {}).values():
            m.reset_states()


class SmallRAFT(RAFT):
    """reference model.py:173-226."""

    variant = 'small'

    def __init__(self, drop_rate=0, iters=12, iters_pred=24, **kwargs):
        super().__init__(drop_rate, iters, iters_pred, **kwargs)
        self.hidden_dim = 96
        self.context_dim = 64
        self.corr_levels = 4
        self.corr_radius = 3

    def _build(self, weights):
        self.fnet = SmallEncoder(output_dim=128, norm_type='instance', drop_rate=self.drop_rate,
                                 weights=weights, prefix='fnet')
        self.cnet = SmallEncoder(output_dim=96 + 64, norm_type=None, drop_rate=self.drop_rate,
                                 weights=weights, prefix='cnet')
        self.update_block = SmallUpdateBlock(filters=96, weights=weights, prefix='update_block')

    def _prepare(self, cnet, st):
        check(_dev.lib().raft_prepare_state_small_f32(_dev.ptr(cnet), st.B, st.h, st.w, C.byref(st.c),
                                                      _dev.stream_ptr()), 'prepare_state_small')

    def _iterate(self, corr, st, iters, flow_up):
        check(_dev.lib().raft_iterate_small_f32(C.byref(self.update_block.c), _dev.ptr(corr._pyr), corr._off,
                                                st.B, st.h, st.w, iters, C.byref(st.c), _dev.ptr(flow_up),
                                                _dev.stream_ptr()), 'iterate_small')

    def _upsample_into(self, st, out):
        check(_dev.lib().raft_upflow8_f32(_dev.ptr(st.flow), st.B, st.h, st.w, _dev.ptr(out),
                                          _dev.stream_ptr()), 'upflow8')

    def upsample_flow(self, flow, mask=None):
        return upflow8(flow)

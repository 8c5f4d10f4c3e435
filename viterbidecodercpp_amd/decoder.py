"""Host-side mirror of the reference's decoder interface, over the C ABI (include/vit_hip.h).

Names, argument meaning and call pattern follow the reference so that tests read like its own:

    table  = ViterbiBranchTable(K, R, G, soft_decision_high, soft_decision_low, soft_dtype)   # viterbi_branch_table.h:34
    vitdec = ViterbiDecoder_Core(table, config)                                               # viterbi_decoder_core.h:170
    vitdec.set_traceback_length(total_input_bits); vitdec.reset()                             # :180, :202
    acc = ViterbiDecoder_HIP.update(vitdec, symbols)                                          # viterbi_decoder_scalar.h:29
    err = acc + vitdec.get_error(); data = vitdec.chainback(total_input_bits)                 # core.h:195, :214

`ViterbiDecoder_HIP` is the GPU decoder strategy (the counterpart of ViterbiDecoder_Scalar / _AVX_u16 ...): update() and
chainback() run as HIP kernels on gfx950; nothing here decodes on the CPU.  `BatchDecoder` is the throughput route:
many independent frames, device-resident tensors, explicit stream.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .codes import DecoderConfig


def _parity(x: np.ndarray) -> np.ndarray:
    x = x.astype(np.uint32)
    x ^= x >> 16
    x ^= x >> 8
    x ^= x >> 4
    x ^= x >> 2
    x ^= x >> 1
    return (x & 1).astype(np.uint8)


class ViterbiBranchTable:
    """Expected symbol value for each half-state and polynomial: bt[i][s] = parity((s << 1) & G[i]) ? high : low
    (include/viterbi/viterbi_branch_table.h:44-54).  Shareable between decoders (README.md:14)."""

    def __init__(self, K: int, R: int, G, soft_decision_high: int, soft_decision_low: int, soft_dtype=np.int16):
        if K < 2 or R < 1:
            raise ValueError("K >= 2 and R >= 1 required")
        if len(G) != R:
            raise ValueError("need exactly R polynomials")
        if not soft_decision_high > soft_decision_low:
            raise ValueError("soft_decision_high must exceed soft_decision_low")
        self.K, self.R, self.G = K, R, tuple(int(g) for g in G)
        self.soft_decision_high, self.soft_decision_low = int(soft_decision_high), int(soft_decision_low)
        self.soft_dtype = np.dtype(soft_dtype)
        self.NUMSTATES = max((1 << (K - 1)) // 2, 1)
        s = np.arange(self.NUMSTATES, dtype=np.uint32) << 1
        rows = [np.where(_parity(s & np.uint32(g)) != 0, self.soft_decision_high, self.soft_decision_low) for g in self.G]
        self._table = np.ascontiguousarray(np.stack(rows).astype(self.soft_dtype))

    def __getitem__(self, index):
        return self._table[index]

    def data(self) -> np.ndarray:
        return self._table


@dataclass
class ViterbiDecoder_Config:
    """include/viterbi/viterbi_decoder_config.h:11-18 (+ the error_t width)."""
    soft_decision_max_error: int
    initial_start_error: int
    initial_non_start_error: int
    renormalisation_threshold: int
    error_dtype: object = np.uint16

    @classmethod
    def from_decoder_config(cls, c: DecoderConfig):
        return cls(c.soft_decision_max_error, c.initial_start_error, c.initial_non_start_error,
                   c.renormalisation_threshold, c.error_dtype)

    def as_array(self):
        return np.asarray([self.soft_decision_max_error, self.initial_start_error, self.initial_non_start_error,
                           self.renormalisation_threshold], dtype=self.error_dtype)


class _Handle:
    def __init__(self, table: ViterbiBranchTable, config: ViterbiDecoder_Config, device: int = 0, blob: bytes = None):
        L = _lib.load()
        self._h = C.c_void_p()
        if blob is not None:
            buf = (C.c_uint8 * len(blob)).from_buffer_copy(blob)
            _lib.check(L.vit_hip_create_from_blob(buf, len(blob), device, C.byref(self._h)))
        else:
            cfg = config.as_array()
            edt = np.dtype(config.error_dtype)
            if (table.soft_dtype.itemsize, edt.itemsize) not in ((2, 2), (1, 1)):
                raise ValueError("(soft_t, error_t) must be (int16, uint16) or (int8, uint8)")
            _lib.check(L.vit_hip_create(table.K, table.R, table.soft_dtype.itemsize, edt.itemsize,
                                        table.data().ctypes.data_as(C.c_void_p), cfg.ctypes.data_as(C.c_void_p), device,
                                        C.byref(self._h)))
        self.info = _lib.VitHipInfo()
        _lib.check(L.vit_hip_get_info(self._h, C.byref(self.info)))

    def refresh(self):
        _lib.check(_lib.load().vit_hip_get_info(self._h, C.byref(self.info)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.load().vit_hip_destroy(self._h)
                self._h = None
        except Exception:
            pass


def pack_blob(table: ViterbiBranchTable, config: ViterbiDecoder_Config) -> bytes:
    """table + config as the byte blob rank 0 broadcasts to the other GPUs (vit_hip_pack_blob)."""
    L = _lib.load()
    edt = np.dtype(config.error_dtype)
    n = L.vit_hip_blob_bytes(table.K, table.R, table.soft_dtype.itemsize, edt.itemsize)
    buf = (C.c_uint8 * n)()
    cfg = config.as_array()
    _lib.check(L.vit_hip_pack_blob(table.K, table.R, table.soft_dtype.itemsize, edt.itemsize,
                                   table.data().ctypes.data_as(C.c_void_p), cfg.ctypes.data_as(C.c_void_p), buf, n))
    return bytes(buf)


class ViterbiDecoder_Core:
    """Decoder state with the reference's public surface (viterbi_decoder_core.h:157-243): m_metrics, m_decisions and
    m_current_decoded_bit are host-visible; update() and chainback() execute on the GPU.

    Streamed update() calls (a few trellis steps each: examples/helpers/puncture_code_helpers.h:51 hands over ONE step per
    call) are queued on the host and run in one launch when the queue is full, when the cursor reaches the end of the traceback
    buffer, or when anything reads the state (m_metrics, m_decisions, get_error, chainback) -- the same scheme as the C++
    header include/viterbi_hip/viterbi_decoder_core.h ("deferred streaming")."""

    MAX_PENDING_STEPS = 2048
    MAX_DEFERRED_CALL_STEPS = 64

    def __init__(self, branch_table: ViterbiBranchTable, config: ViterbiDecoder_Config, device: int = 0):
        self.m_branch_table = branch_table
        self.m_config = config
        self.K, self.R = branch_table.K, branch_table.R
        self.TOTAL_STATE_BITS = self.K - 1
        self.NUMSTATES = 1 << (self.K - 1)
        self.TOTAL_BLOCKS = max(self.NUMSTATES // 64, 1)
        self._handle = _Handle(branch_table, config, device)
        self._metrics = np.zeros(self.NUMSTATES, dtype=config.error_dtype)
        self._decisions = np.zeros((0, self.TOTAL_BLOCKS), dtype=np.uint64)
        self._pending, self._pending_steps, self._unreported = [], 0, 0
        self._dropped = 0
        self._dev_begin = self._dev_end = 0     # rows [begin, end) of _decisions are stale here: they live in the handle's device row store
        self._exact_update_return = False
        self.m_current_decoded_bit = 0
        self.reset()
        self.set_traceback_length(0)

    # the public data members of the reference: reading them brings the decoder up to date first
    @property
    def m_metrics(self):
        self.flush_pending()
        return self._metrics

    @property
    def m_decisions(self):
        # the array handed out is writable: queued steps run, device-resident rows come home, the host copy is the authority again
        self.flush_pending()
        self.fetch_device_rows()
        return self._decisions

    def fetch_device_rows(self):
        """update() leaves the rows it computes in the handle's device row store (vit_hip_update_host_lazy); this copies the stale
        range back and makes the host rows authoritative (what reading m_decisions does)."""
        b, e = self._dev_begin, min(self._dev_end, len(self._decisions))
        self._dev_begin = self._dev_end = 0
        if e > b:
            rows = np.zeros((e - b, self.TOTAL_BLOCKS), dtype=np.uint64)
            _lib.check(_lib.load().vit_hip_fetch_decisions_host(self._handle._h, b, e - b, rows.ctypes.data_as(C.c_void_p)))
            self._decisions[b:e] = rows

    @property
    def rows_on_device(self) -> int:
        return self._dev_end - self._dev_begin

    def set_traceback_length(self, traceback_length: int):
        self.flush_pending()
        self.fetch_device_rows()
        new_length = traceback_length + self.TOTAL_STATE_BITS
        old = self._decisions
        self._decisions = np.zeros((new_length, self.TOTAL_BLOCKS), dtype=np.uint64)
        n = min(len(old), new_length)
        self._decisions[:n] = old[:n]
        if self.m_current_decoded_bit > new_length:
            self.m_current_decoded_bit = new_length

    def get_traceback_length(self) -> int:
        return len(self._decisions) - self.TOTAL_STATE_BITS

    def get_error(self, end_state: int = 0) -> int:
        assert end_state < self.NUMSTATES
        return int(self.m_metrics[end_state])

    def set_exact_update_return(self, exact: bool):
        """True: nothing is deferred -- every update() call runs at once and returns what the reference's update() returns for
        that call (one GPU launch per call); False (default): short calls are queued (see the class docstring)."""
        self.flush_pending()
        self._exact_update_return = bool(exact)

    def reset(self, starting_state: int = 0):
        # queued steps are run before the state is dropped (the reference has run them); a renormalisation sum no update() call has
        # returned yet is DROPPED: a frame's tail never inflates the next frame's total (collect it with
        # take_unreported_renormalisation() before reset(), or stream in exact mode)
        self.flush_pending()
        self._dropped, self._unreported = self._unreported, 0
        self.m_current_decoded_bit = 0
        self._metrics[:] = self.m_config.initial_non_start_error
        self._metrics[starting_state & (self.NUMSTATES - 1)] = self.m_config.initial_start_error

    def _run(self, symbols, steps, first_row):
        """one launch (vit_hip_update_host_lazy): the rows stay on the device; a call that completes the frame also chains it back for
        (traceback length, end state 0), so that the chainback() behind it costs a memcpy"""
        if self._dev_end > self._dev_begin and (first_row > self._dev_end or first_row + steps < self._dev_begin):
            self.fetch_device_rows()             # not adjacent to the device-resident range: one range only
        rs = C.c_uint64(0)
        _lib.check(_lib.load().vit_hip_update_host_lazy(self._handle._h, self._metrics.ctypes.data_as(C.c_void_p),
                                                        symbols.ctypes.data_as(C.c_void_p), steps, first_row,
                                                        self.get_traceback_length(), 0, C.byref(rs)))
        if self._dev_end == self._dev_begin:
            self._dev_begin, self._dev_end = first_row, first_row + steps
        else:
            self._dev_begin, self._dev_end = min(self._dev_begin, first_row), max(self._dev_end, first_row + steps)
        return int(rs.value)

    def enqueue_steps(self, symbols, steps):
        self._pending.append(symbols)
        self._pending_steps += steps
        self.m_current_decoded_bit += steps
        if self._exact_update_return or self._pending_steps >= self.MAX_PENDING_STEPS or self.m_current_decoded_bit >= len(self._decisions):
            self.flush_pending()

    def flush_pending(self):
        if not self._pending_steps:
            return
        steps, self._pending_steps = self._pending_steps, 0
        symbols = np.ascontiguousarray(np.concatenate(self._pending))
        self._pending = []
        self._unreported += self._run(symbols, steps, self.m_current_decoded_bit - steps)

    def take_unreported_renormalisation(self) -> int:
        v, self._unreported = self._unreported, 0
        return v

    def last_dropped_renormalisation(self) -> int:
        """what the last reset() dropped: a renormalisation sum that had been computed (by a flush) but that no update() call had
        returned and nobody had collected -- 0 unless a frame shorter than the traceback buffer was streamed in short calls"""
        return self._dropped

    def chainback(self, total_bits: int, end_state: int = 0) -> np.ndarray:
        assert self.get_traceback_length() >= total_bits
        assert self.m_current_decoded_bit - self.TOTAL_STATE_BITS >= total_bits
        assert end_state < self.NUMSTATES
        out = np.zeros((total_bits + 7) // 8, dtype=np.uint8)
        self.flush_pending()
        if self._dev_begin == 0 and self._dev_end >= total_bits + self.TOTAL_STATE_BITS:
            # every row the traceback reads is still on the device: nothing travels, and if the update that completed the frame
            # already decoded exactly these bits no kernel runs either
            _lib.check(_lib.load().vit_hip_chainback_host_lazy(self._handle._h, total_bits, end_state, out.ctypes.data_as(C.c_void_p)))
            return out
        rows = np.ascontiguousarray(self.m_decisions[:total_bits + self.TOTAL_STATE_BITS])
        _lib.check(_lib.load().vit_hip_chainback_host(self._handle._h, rows.ctypes.data_as(C.c_void_p), total_bits,
                                                      end_state, out.ctypes.data_as(C.c_void_p)))
        return out


class ViterbiDecoder_HIP:
    """The MI355X decoder strategy: same concept as ViterbiDecoder_Scalar (static update(), is_valid)."""
    is_valid = True

    @staticmethod
    def update(base: ViterbiDecoder_Core, symbols) -> int:
        """returns the renormalisation sum of the steps COMPUTED since the last value it returned (zero while short calls are
        queued, their whole sum from the call that runs them): the caller's running total is the reference's whenever nothing is
        queued -- in particular after the call that completes a frame."""
        symbols = np.ascontiguousarray(symbols, dtype=base.m_branch_table.soft_dtype).reshape(-1)
        N = symbols.size
        assert N % base.R == 0
        total_decoded_bits = N // base.R
        max_decoded_bits = base.get_traceback_length() + base.TOTAL_STATE_BITS
        assert total_decoded_bits + base.m_current_decoded_bit <= max_decoded_bits
        if total_decoded_bits == 0:
            return 0
        if total_decoded_bits <= base.MAX_DEFERRED_CALL_STEPS:
            base.enqueue_steps(symbols.copy(), total_decoded_bits)
            return base.take_unreported_renormalisation()
        base.flush_pending()
        rs = base._run(symbols, total_decoded_bits, base.m_current_decoded_bit)
        base.m_current_decoded_bit += total_decoded_bits
        return rs + base.take_unreported_renormalisation()


class BatchDecoder:
    """Frame-parallel decoder on device-resident torch tensors (vit_hip_*_batch).  torch is plumbing only: it owns the
    HBM buffers and the stream."""

    def __init__(self, branch_table: ViterbiBranchTable = None, config: ViterbiDecoder_Config = None, device=0,
                 blob: bytes = None, plan: int = _lib.PLAN_AUTO):
        import torch

        self.torch = torch
        self.device_index = device if isinstance(device, int) else torch.device(device).index or 0
        self.device = torch.device("cuda", self.device_index)
        self._handle = _Handle(branch_table, config, self.device_index, blob=blob)
        if plan != _lib.PLAN_AUTO:
            self.set_plan(plan)
        i = self._handle.info
        self.K, self.R, self.N, self.W = i.K, i.R, i.num_states, i.decision_words
        self.soft_bytes, self.error_bytes = i.soft_bytes, i.error_bytes
        self._ws = None

    @property
    def plan(self) -> int:
        return self._handle.info.plan

    def set_plan(self, plan: int):
        _lib.check(_lib.load().vit_hip_set_plan(self._handle._h, plan))
        self._handle.refresh()

    @property
    def plan_note(self) -> str:
        """vit_hip_plan_note: which plan runs and, where that is the slow compatibility plan, whether a faster one exists"""
        return _lib.load().vit_hip_plan_note(self._handle._h).decode()

    def workspace_bytes(self, frames: int, L: int) -> int:
        return _lib.load().vit_hip_workspace_bytes(self._handle._h, frames, L)

    def _workspace(self, frames, L, workspace=None):
        need = self.workspace_bytes(frames, L)
        if workspace is not None:
            if workspace.numel() * workspace.element_size() < need or workspace.data_ptr() % 256 != 0:
                raise ValueError("workspace too small or not 256-byte aligned")
            return workspace
        if self._ws is None or self._ws.numel() < need:
            self._ws = None
            self._ws = self.torch.empty(need, dtype=self.torch.uint8, device=self.device)
        assert self._ws.data_ptr() % 256 == 0
        return self._ws

    def kernel_resources(self, kernel: int = 0) -> dict:
        """what one wave / workgroup of this decoder's kernel allocates, from its kernel descriptor (vit_hip_get_kernel_resources;
        kernel: _lib.KERNEL_UPDATE / _CHAINBACK / _CHAINBACK_ALT / _RESUME)"""
        r = _lib.VitHipKernelResources()
        _lib.check(_lib.load().vit_hip_get_kernel_resources(self._handle._h, int(kernel), C.byref(r)))
        return _lib._resources_dict(r)

    def new_workspace(self, frames: int, L: int):
        """a private decision workspace (for double-buffered pipelines: update of batch i+1 beside chainback of batch i)."""
        return self.torch.empty(self.workspace_bytes(frames, L), dtype=self.torch.uint8, device=self.device)

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def _check_symbols(self, symbols, n_steps):
        t = self.torch
        want = t.int16 if self.soft_bytes == 2 else t.int8
        if symbols.dtype != want or not symbols.is_cuda or not symbols.is_contiguous():
            raise ValueError(f"symbols must be a contiguous {want} CUDA tensor")
        frames = symbols.shape[0]
        if symbols.numel() != frames * n_steps * self.R:
            raise ValueError("symbols must have shape [frames][n_steps][R]")
        return frames

    def update(self, symbols, L: int, n_steps: int = None, start_state=None, want_metrics=True, metrics_out=None,
               renorm_out=None, workspace=None):
        """reset + update over n_steps (default L+K-1) steps.  returns (final_metrics [F][N], renorm_sum [F]);
        with want_metrics=False and no *_out buffers nothing but the decision workspace is written."""
        t = self.torch
        n_steps = (L + self.K - 1) if n_steps is None else n_steps
        frames = self._check_symbols(symbols, n_steps)
        ws = self._workspace(frames, L, workspace)
        edt = t.int16 if self.error_bytes == 2 else t.uint8  # int16 carries the uint16 bit pattern
        met, rs = metrics_out, renorm_out
        if want_metrics:
            met = t.empty((frames, self.N), dtype=edt, device=self.device) if met is None else met
            rs = t.empty(frames, dtype=t.int64, device=self.device) if rs is None else rs
        ss = None
        if start_state is not None:
            ss = t.as_tensor(start_state, dtype=t.int32, device=self.device).contiguous()
        _lib.check(_lib.load().vit_hip_update_batch(
            self._handle._h, C.c_void_p(symbols.data_ptr()), frames, n_steps, L, C.c_void_p(ws.data_ptr()),
            ws.numel() * ws.element_size(),
            C.c_void_p(met.data_ptr()) if met is not None else None, C.c_void_p(rs.data_ptr()) if rs is not None else None,
            C.c_void_p(ss.data_ptr()) if ss is not None else None, self._stream()))
        return met, rs

    def reset_batch(self, frames: int, start_state=None, metrics_out=None):
        """ViterbiDecoder_Core::reset for every frame (core.h:202-211): the device-resident metrics [F][N] a streamed decode
        starts from."""
        t = self.torch
        edt = t.int16 if self.error_bytes == 2 else t.uint8
        met = t.empty((frames, self.N), dtype=edt, device=self.device) if metrics_out is None else metrics_out
        ss = None
        if start_state is not None:
            ss = t.as_tensor(start_state, dtype=t.int32, device=self.device).contiguous()
        _lib.check(_lib.load().vit_hip_reset_batch(self._handle._h, frames, C.c_void_p(ss.data_ptr()) if ss is not None else None,
                                                   C.c_void_p(met.data_ptr()), self._stream()))
        return met

    def update_resume(self, symbols, L: int, first_step: int, metrics, n_steps: int = None, symbol_frame_stride: int = 0,
                      renorm_out=None, workspace=None):
        """update() on decoders that already hold state (no reset): consumes `n_steps` more trellis steps of every frame from
        cursor `first_step`, metrics [F][N] updated in place, decision rows [first_step, first_step + n_steps) written.
        `symbols` holds each frame's chunk [n_steps][R] -- either a [F][n_steps][R] tensor, or any tensor whose element
        `f * symbol_frame_stride` starts frame f's chunk (e.g. a view into the whole [F][S][R] batch).  returns renorm_sum [F]
        for THIS call."""
        t = self.torch
        frames = metrics.shape[0]
        if n_steps is None:
            n_steps = symbols.shape[1]
        want = t.int16 if self.soft_bytes == 2 else t.int8
        if symbols.dtype != want or not symbols.is_cuda:
            raise ValueError(f"symbols must be a {want} CUDA tensor")
        if symbol_frame_stride == 0 and not symbols.is_contiguous():
            raise ValueError("packed chunks must be contiguous (or pass symbol_frame_stride)")
        if metrics.dim() != 2 or metrics.shape[1] != self.N or not metrics.is_contiguous():
            raise ValueError("metrics must be a contiguous [frames][N] tensor")
        if symbol_frame_stride == 0 and symbols.dim() == 3 and (symbols.shape[0] != frames or symbols.shape[2] != self.R):
            raise ValueError("packed chunks must have shape [frames][n_steps][R]")
        # the kernels read frame f's chunk at f * stride: the tensor must cover the last frame's chunk
        stride = symbol_frame_stride or n_steps * self.R
        if frames and symbols.numel() < (frames - 1) * stride + n_steps * self.R:
            raise ValueError(f"symbols holds {symbols.numel()} elements; {frames} chunks of {n_steps} x {self.R} at stride "
                             f"{stride} need {(frames - 1) * stride + n_steps * self.R}")
        ws = self._workspace(frames, L, workspace)
        rs = t.empty(frames, dtype=t.int64, device=self.device) if renorm_out is None else renorm_out
        _lib.check(_lib.load().vit_hip_update_batch_resume(
            self._handle._h, C.c_void_p(symbols.data_ptr()), symbol_frame_stride, frames, first_step, n_steps, L,
            C.c_void_p(ws.data_ptr()), ws.numel() * ws.element_size(), C.c_void_p(metrics.data_ptr()),
            C.c_void_p(rs.data_ptr()), self._stream()))
        return rs

    def chainback(self, frames: int, L: int, end_state=None, out=None, workspace=None, kernel=None):
        """kernel: None (the code's stand-alone chainback kernel) or _lib.KERNEL_CHAINBACK / _lib.KERNEL_CHAINBACK_ALT
        (vit_hip_chainback_batch_ex: the other chainback kernel of the K = 7 / 9 register-plan codes; same results)"""
        t = self.torch
        ws = self._workspace(frames, L, workspace)
        if out is None:
            out = t.empty((frames, (L + 7) // 8), dtype=t.uint8, device=self.device)
        es = None
        if end_state is not None:
            es = t.as_tensor(end_state, dtype=t.int32, device=self.device).contiguous()
        esp = C.c_void_p(es.data_ptr()) if es is not None else None
        if kernel is None:
            _lib.check(_lib.load().vit_hip_chainback_batch(self._handle._h, C.c_void_p(ws.data_ptr()), frames, L,
                                                           C.c_void_p(out.data_ptr()), esp, self._stream()))
        else:
            _lib.check(_lib.load().vit_hip_chainback_batch_ex(self._handle._h, C.c_void_p(ws.data_ptr()), frames, L,
                                                              C.c_void_p(out.data_ptr()), esp, self._stream(), int(kernel)))
        return out

    def decode(self, symbols, L: int, out=None, want_metrics=False):
        """reset -> update -> chainback for every frame; returns bytes [F][L/8] (+ metrics, renorm when asked)."""
        t = self.torch
        frames = self._check_symbols(symbols, L + self.K - 1)
        ws = self._workspace(frames, L)
        if out is None:
            out = t.empty((frames, (L + 7) // 8), dtype=t.uint8, device=self.device)
        edt = t.int16 if self.error_bytes == 2 else t.uint8  # int16 carries the uint16 bit pattern
        met = t.empty((frames, self.N), dtype=edt, device=self.device) if want_metrics else None
        rs = t.empty(frames, dtype=t.int64, device=self.device) if want_metrics else None
        _lib.check(_lib.load().vit_hip_decode_batch(
            self._handle._h, C.c_void_p(symbols.data_ptr()), frames, L, C.c_void_p(ws.data_ptr()), ws.numel(),
            C.c_void_p(out.data_ptr()), C.c_void_p(met.data_ptr()) if met is not None else None,
            C.c_void_p(rs.data_ptr()) if rs is not None else None, None, self._stream()))
        return (out, met, rs) if want_metrics else out

    def depuncture(self, punctured, mask, out=None):
        """Depuncturing front-end (examples/helpers/puncture_code_helpers.h:17-55) for a batch: `punctured` is the device
        tensor [F][P] of transmitted symbols, `mask` the puncturing vector over one whole frame (truthy = transmitted, one
        entry per mother-code symbol, S*R in all); returns [F][S][R] with 0 (erasure) at the punctured positions."""
        t = self.torch
        mask = np.asarray(mask).astype(bool).reshape(-1)
        if mask.size % self.R != 0:
            raise ValueError("the puncturing vector must cover whole trellis steps (a multiple of R symbols)")
        if not (punctured.is_cuda and punctured.dim() == 2 and punctured.is_contiguous()):
            raise ValueError("punctured symbols must be a contiguous [frames][P] device tensor")
        sdt = t.int16 if self.soft_bytes == 2 else t.int8
        if punctured.dtype != sdt or punctured.shape[1] != int(mask.sum()):
            raise ValueError("punctured symbols: wrong dtype or count for this puncturing vector")
        key = mask.tobytes()
        cached = getattr(self, "_depuncture_map", None)
        if cached is None or cached[0] != key:             # the map is per puncturing scheme: build and upload it once
            idx = np.where(mask, np.cumsum(mask) - 1, -1).astype(np.int32)
            cached = (key, t.from_numpy(idx).to(self.device))
            self._depuncture_map = cached
        d_idx = cached[1]
        frames = punctured.shape[0]
        if out is None:
            out = t.empty((frames, mask.size // self.R, self.R), dtype=sdt, device=self.device)
        _lib.check(_lib.load().vit_hip_depuncture_batch(self._handle._h, C.c_void_p(punctured.data_ptr()), punctured.shape[1],
                                                        C.c_void_p(d_idx.data_ptr()), mask.size, frames,
                                                        C.c_void_p(out.data_ptr()), self._stream()))
        return out

    def synth(self, frames: int, L: int, ebn0_db, seed: int = 1, first_frame: int = 0, tx_out=None, symbols_out=None):
        """synthetic AWGN frames of this decoder's code, generated in HBM by one HIP kernel (vit_hip_synth_batch: the
        reference BER harness's generator, examples/run_snr_ber.cpp:311-359).  ebn0_db=None: noise-free symbols at exactly
        high / low.  returns (tx_bytes [F][L/8] uint8, symbols [F][L+K-1][R] soft dtype), both on the device."""
        t = self.torch
        sdt = t.int16 if self.soft_bytes == 2 else t.int8
        tx = t.empty((frames, L // 8), dtype=t.uint8, device=self.device) if tx_out is None else tx_out
        sym = t.empty((frames, L + self.K - 1, self.R), dtype=sdt, device=self.device) if symbols_out is None else symbols_out
        _lib.check(_lib.load().vit_hip_synth_batch(self._handle._h, frames, L, int(seed), int(first_frame),
                                                   0.0 if ebn0_db is None else float(ebn0_db), 1 if ebn0_db is None else 0,
                                                   C.c_void_p(tx.data_ptr()), C.c_void_p(sym.data_ptr()), self._stream()))
        return tx, sym

    def count_bit_errors(self, a, b, count=None):
        """number of differing bits between two device byte tensors (get_total_bit_errors, test_helpers.h:95-104), added to
        the one-element int64 device tensor `count` (created zeroed when None); no host synchronisation."""
        t = self.torch
        if a.dtype != t.uint8 or b.dtype != t.uint8 or a.numel() != b.numel() or not (a.is_contiguous() and b.is_contiguous()):
            raise ValueError("need two contiguous uint8 tensors of equal size")
        if count is None:
            count = t.zeros(1, dtype=t.int64, device=self.device)
        _lib.check(_lib.load().vit_hip_count_bit_errors(self._handle._h, C.c_void_p(a.data_ptr()), C.c_void_p(b.data_ptr()),
                                                        a.numel(), C.c_void_p(count.data_ptr()), self._stream()))
        return count

    def export_decisions(self, frames: int, L: int, n_steps: int = None, workspace=None, first_frame: int = 0):
        """decision history in the reference layout: int64 tensor [F][n_steps][W] (bit pattern of uint64 words).  The
        workspace is an array of independent slabs of `workspace_tile_frames` frames (vit_hip_info), so `frames` frames
        starting at a slab-aligned `first_frame` of a larger batch can be exported on their own."""
        t = self.torch
        n_steps = (L + self.K - 1) if n_steps is None else n_steps
        if first_frame:
            self._handle.refresh()
            tile = self._handle.info.workspace_tile_frames
            if first_frame % tile:
                raise ValueError(f"first_frame must be a multiple of {tile} for this plan")
            base = self._ws if workspace is None else workspace
            workspace = base[(first_frame // tile) * _lib.load().vit_hip_workspace_slab_bytes(self._handle._h, L):]
            if workspace.numel() * workspace.element_size() < self.workspace_bytes(frames, L) - 255:
                raise ValueError("workspace too small for that sub-range")
            ws = workspace                   # a slab address: 256-byte alignment is only asked of whole workspaces
        else:
            ws = self._workspace(frames, L, workspace)
        dec = t.empty((frames, n_steps, self.W), dtype=t.int64, device=self.device)
        _lib.check(_lib.load().vit_hip_export_decisions(self._handle._h, C.c_void_p(ws.data_ptr()), frames, n_steps, L,
                                                        C.c_void_p(dec.data_ptr()), self._stream()))
        return dec


class DecodePipeline:
    """vit_hip_pipeline_*: a stream of batches through one decoder, scheduled by the library (chainback of batch i beside the
    update of batch i+1, two updates in flight for small batches -- include/vit_hip.h).  The Python face of the C++
    ViterbiDecoder_HIP_Pipeline (include/viterbi_hip/viterbi_decoder_hip_batch.h); the loop it replaces is the reference
    benchmark's per-frame reset -> update -> chainback (examples/run_benchmark.cpp:266-282)."""

    def __init__(self, decoder: BatchDecoder, max_frames: int, L: int, options: "_lib.VitHipPipelineOptions" = None):
        """options: a _lib.VitHipPipelineOptions to override the library's schedule rules (vit_hip_pipeline_create_ex); None: the rules"""
        self.decoder, self.max_frames, self.L = decoder, int(max_frames), int(L)
        self._p = C.c_void_p()
        with decoder.torch.cuda.device(decoder.device):
            if options is None:
                _lib.check(_lib.load().vit_hip_pipeline_create(decoder._handle._h, self.max_frames, self.L, C.byref(self._p)))
            else:
                _lib.check(_lib.load().vit_hip_pipeline_create_ex(decoder._handle._h, self.max_frames, self.L, C.byref(options),
                                                                  C.byref(self._p)))
        self.schedule = _lib.VitHipPipelineSchedule()
        _lib.check(_lib.load().vit_hip_pipeline_get_schedule_v2(self._p, C.byref(self.schedule), C.sizeof(self.schedule)))
        # the pipeline runs on private non-blocking streams: these two events order it against torch's streams
        t = decoder.torch
        with t.cuda.device(decoder.device):
            self._inputs_ready, self._done = t.cuda.Event(), t.cuda.Event()
            self._done.record()                                  # materialises the hipEvent_t the pipeline re-records

    def submit(self, symbols, out, end_state=None, after_current_stream=True):
        """enqueue one batch: symbols [F][L+K-1][R] -> out [F][L/8]; both must stay untouched until sync() / wait_done().
        after_current_stream (default): the batch's update kernel is ordered behind everything already enqueued on torch's
        current stream of the decoder's device (an event recorded there, vit_hip_pipeline_wait_event), so
        `sym = dec.synth(...); pipe.submit(sym, out)` needs no synchronisation in between.  False: the caller guarantees the
        inputs are complete (what a C host without a producer stream does)."""
        t = self.decoder.torch
        frames = self.decoder._check_symbols(symbols, self.L + self.decoder.K - 1)
        if out.dtype != t.uint8 or not out.is_contiguous() or out.numel() != frames * ((self.L + 7) // 8):
            raise ValueError("out must be a contiguous uint8 tensor [frames][ceil(L/8)]")
        L = _lib.load()
        if after_current_stream:
            self._inputs_ready.record(t.cuda.current_stream(self.decoder.device))
            _lib.check(L.vit_hip_pipeline_wait_event(self._p, C.c_void_p(self._inputs_ready.cuda_event)))
        _lib.check(L.vit_hip_pipeline_submit(self._p, C.c_void_p(symbols.data_ptr()), frames, C.c_void_p(out.data_ptr()),
                                             C.c_void_p(end_state.data_ptr()) if end_state is not None else None,
                                             C.c_void_p(self._done.cuda_event)))

    def wait_done(self, stream=None):
        """order `stream` (default: torch's current stream) behind every batch submitted so far, without blocking the host:
        torch work enqueued on it afterwards may read the outputs.  (A later submit(after_current_stream=True) from the same
        stream is then ordered behind these batches too -- call it only where the outputs are consumed.)"""
        t = self.decoder.torch
        (stream or t.cuda.current_stream(self.decoder.device)).wait_event(self._done)

    def sync(self):
        _lib.check(_lib.load().vit_hip_pipeline_sync(self._p))

    def export_last_decisions(self, frames: int = None, n_steps: int = None):
        """decision rows of the most recently submitted (sub-)batch, after sync(): returns (first_frame, rows) where rows
        [n][n_steps][W] belong to frames first_frame .. first_frame + n - 1 of the submitted batch (n = `frames` or the whole
        sub-batch)"""
        dec, t = self.decoder, self.decoder.torch
        n_steps = (self.L + dec.K - 1) if n_steps is None else n_steps
        ws, f0, nf = C.c_void_p(), C.c_size_t(0), C.c_size_t(0)
        _lib.check(_lib.load().vit_hip_pipeline_last_workspace(self._p, C.byref(ws), C.byref(f0), C.byref(nf)))
        n = nf.value if frames is None else min(frames, nf.value)
        out = t.empty((n, n_steps, dec.W), dtype=t.int64, device=dec.device)
        _lib.check(_lib.load().vit_hip_export_decisions(dec._handle._h, ws, n, n_steps, self.L, C.c_void_p(out.data_ptr()),
                                                        dec._stream()))
        return f0.value, out

    def set_timing(self, enable: bool):
        _lib.check(_lib.load().vit_hip_pipeline_set_timing(self._p, 1 if enable else 0))

    def timing(self):
        """(update_ms, chainback_ms, complete_ms) float32 arrays, one entry per batch completed since set_timing(True)"""
        L = _lib.load()
        n = C.c_size_t(0)
        _lib.check(L.vit_hip_pipeline_get_timing(self._p, 0, None, None, None, C.byref(n)))
        u, c, d = (np.zeros(n.value, dtype=np.float32) for _ in range(3))
        _lib.check(L.vit_hip_pipeline_get_timing(self._p, n.value, u.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p),
                                                 d.ctypes.data_as(C.c_void_p), C.byref(n)))
        return u, c, d

    def close(self):
        if getattr(self, "_p", None):
            _lib.load().vit_hip_pipeline_destroy(self._p)
            self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

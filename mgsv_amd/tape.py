"""Launch tape: record the libmade_hip launches of one step once, replay them from one C loop (include/made_hip.h: made_tape_*).

    with LaunchTape.record() as tape:      # the step runs normally (and is recorded)
        trainer.train_step(...)
    tape.replay()                          # re-issues the same launches on the same streams

While recording, torch's stream dependencies (Stream.wait_stream / wait_event, Event.record) are mirrored into the tape, every
tensor handed to a library call is kept alive (the tape holds raw pointers), and fills / contiguous copies go through `zero_` /
`copy_` below.  Any OTHER device work of the framework inside the recorded region (an ATen kernel) would be missing from the replay:
`LaunchTape.record(check=True)` (the default) watches the dispatcher while the step is recorded and raises `ForeignKernelError` naming every
framework operator that touched a device tensor (allocations and views aside) -- a configuration whose step still contains such an
operator cannot be replayed from the tape and is refused instead of silently skipping that work.
"""
from __future__ import annotations

import contextlib
import ctypes as C
from typing import List, Optional

import torch

from . import _lib

Tensor = torch.Tensor
_recording: Optional["LaunchTape"] = None
_deferred: list = []          # (handle, kept tensors) of tapes whose finaliser ran where a synchronize was not allowed (LaunchTape.__del__)


def _release_deferred() -> None:
    """Frees what finalisers parked: behind a device synchronize, and only where one is allowed (no capture, no recording)."""
    if not _deferred:
        return
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        if _recording is not None or torch.cuda.is_current_stream_capturing():
            return
        torch.cuda.synchronize()
    while _deferred:
        handle, keep = _deferred.pop()
        try:
            _lib.lib().made_tape_free(handle)
        except Exception:
            pass
        del keep


import atexit as _atexit  # noqa: E402

_atexit.register(_release_deferred)


class ForeignKernelError(_lib.MadeError):
    """framework (ATen) device work inside a recorded step: it would be missing from every replay"""


# operators that launch nothing: allocation, views, metadata, host reads of host tensors
_NO_KERNEL = {"empty", "empty_like", "empty_strided", "new_empty", "new_empty_strided", "view", "_unsafe_view", "reshape", "_reshape_alias", "slice", "select",
              "as_strided", "permute", "transpose", "t", "expand", "unsqueeze", "squeeze", "detach", "alias", "unbind", "split", "split_with_sizes",
              "chunk", "narrow", "unfold", "view_as", "contiguous", "lift_fresh", "resolve_conj", "resolve_neg", "size", "stride", "numel", "dim",
              "is_contiguous", "storage_offset", "sym_size", "sym_stride", "sym_numel", "sym_storage_offset", "_to_copy_noop", "view_as_real",
              "is_pinned", "set_", "record_stream", "flatten", "unflatten", "movedim", "swapaxes", "diagonal"}


def _make_watcher(found: list, every_device: bool):
    from torch.utils._python_dispatch import TorchDispatchMode

    class _Watcher(TorchDispatchMode):
        def __torch_dispatch__(self, func, types, args=(), kwargs=None):
            kwargs = kwargs or {}
            out = func(*args, **kwargs)
            name = func.overloadpacket.__name__ if hasattr(func, "overloadpacket") else str(func)
            if name not in _NO_KERNEL and not _in_callback:
                def on_dev(x):
                    return isinstance(x, torch.Tensor) and (x.is_cuda or every_device)
                flat = list(args) + list(kwargs.values()) + (list(out) if isinstance(out, (tuple, list)) else [out])
                flat = [y for x in flat for y in (x if isinstance(x, (tuple, list)) else [x])]
                if any(on_dev(x) for x in flat):
                    found.append(name)
            return out
    return _Watcher()


def recording() -> bool:
    return _recording is not None


def keep(t) -> None:
    """called by ops._p for every tensor a library call receives"""
    if _recording is not None and t is not None:
        _recording._keep.append(t)


def zero_(t: Tensor) -> Tensor:
    """t.zero_() as a recorded stream-ordered fill (made_memset_async); t must be contiguous."""
    assert t.is_contiguous() and t.is_cuda
    keep(t)
    _lib.check(_lib.lib().made_memset_async(t.data_ptr(), 0, t.numel() * t.element_size(), torch.cuda.current_stream().cuda_stream), "made_memset_async")
    return t


def copy_(dst: Tensor, src: Tensor) -> Tensor:
    """dst.copy_(src) for contiguous tensors of one dtype and size, as a recorded device-to-device copy."""
    assert dst.is_contiguous() and src.is_contiguous() and dst.dtype == src.dtype and dst.numel() == src.numel() and dst.is_cuda and src.is_cuda
    keep(dst); keep(src)
    _lib.check(_lib.lib().made_copy_async(dst.data_ptr(), src.data_ptr(), dst.numel() * dst.element_size(), torch.cuda.current_stream().cuda_stream),
               "made_copy_async")
    return dst


_in_callback = False


def callback(fn) -> None:
    """Run fn() now and, while a tape is being recorded, again at this point of the issue order in every replay (made_tape_callback):
    host-side work that belongs BETWEEN the step's launches -- the data-parallel gradient all-reduces (the framework's RCCL calls order
    themselves against the stream that is current when they are made, so the replay makes them under the stream that was current at
    recording time).  What fn does to device tensors is not checked by the recorder's watcher (it IS replayed)."""
    global _in_callback
    tp = _recording
    if tp is not None:
        rec_stream = torch.cuda.current_stream() if torch.cuda.is_available() else None

        def run():
            global _in_callback
            _in_callback = True
            try:
                if rec_stream is not None:
                    with torch.cuda.stream(rec_stream):
                        fn()
                else:
                    fn()
            finally:
                _in_callback = False

        err = tp._cb_err                                      # (the closure holds this small holder, not the tape: tp._keep -> cfn -> closure
                                                              #  would otherwise be a reference cycle only the cyclic GC frees, at a time of its choosing)
        def trampoline(_user):
            try:
                run()
                return 0
            except Exception as ex:                           # (an exception must not cross the C frame: reported through the replay's status)
                err[0] = ex
                return 1
        cfn = _CB_TYPE(trampoline)
        tp._keep.append(cfn)
        _lib.check(_lib.lib().made_tape_callback(cfn, None), "made_tape_callback")
        run()
    else:
        fn()


_CB_TYPE = C.CFUNCTYPE(C.c_int, C.c_void_p)


class LaunchTape:
    def __init__(self):
        self.handle = C.c_uint64(0)
        self._keep: List[object] = []
        self._slots = {}
        self._cb_err: List[Optional[BaseException]] = [None]   # what a replayed host callback raised (see callback())

    @property
    def callback_error(self) -> Optional[BaseException]:
        return self._cb_err[0]

    @callback_error.setter
    def callback_error(self, ex: Optional[BaseException]) -> None:
        self._cb_err[0] = ex

    # ---- recording
    @classmethod
    @contextlib.contextmanager
    def record(cls, check: bool = True, _every_device: bool = False):
        """check: refuse a step that contains framework kernels (see the module docstring); `tape.foreign_ops` lists them either way
        when check is False.  (_every_device: the CPU test of the watcher counts host tensors too.)"""
        global _recording
        assert _recording is None, "tapes do not nest"
        _release_deferred()
        tape = cls()
        tape.foreign_ops = []
        watcher = _make_watcher(tape.foreign_ops, _every_device)
        lib = _lib.lib()
        S, E = torch.cuda.Stream, torch.cuda.Event
        o_ws, o_we, o_rec = S.wait_stream, S.wait_event, E.record

        def slot_of(ev) -> int:
            return tape._slots.setdefault(id(ev), len(tape._slots))

        def wait_stream(self, other):
            o_ws(self, other)
            s = len(tape._slots)
            tape._slots[("ws", s)] = s
            _lib.check(lib.made_tape_event(0, s, other.cuda_stream), "made_tape_event")
            _lib.check(lib.made_tape_event(1, s, self.cuda_stream), "made_tape_event")

        def wait_event(self, ev):
            o_we(self, ev)
            tape._keep.append(ev)
            _lib.check(lib.made_tape_event(1, slot_of(ev), self.cuda_stream), "made_tape_event")

        def record(self, stream=None):
            st = stream if stream is not None else torch.cuda.current_stream()
            o_rec(self, st)
            tape._keep.append(self)
            _lib.check(lib.made_tape_event(0, slot_of(self), st.cuda_stream), "made_tape_event")

        _lib.check(lib.made_tape_begin(), "made_tape_begin")
        _recording = tape
        S.wait_stream, S.wait_event, E.record = wait_stream, wait_event, record
        try:
            with watcher:
                yield tape
        finally:
            S.wait_stream, S.wait_event, E.record = o_ws, o_we, o_rec
            _recording = None
            _lib.check(lib.made_tape_end(C.byref(tape.handle)), "made_tape_end")
        _release_deferred()
        if check and tape.foreign_ops:
            import collections
            c = collections.Counter(tape.foreign_ops)
            tape.close()
            raise ForeignKernelError("the recorded step contains framework kernels a replay would skip: "
                                     + ", ".join(f"aten::{k} x{v}" for k, v in sorted(c.items()))
                                     + " -- this configuration cannot run from the launch tape (use mode='graph' or the eager step)")

    # ---- replay
    def replay(self) -> None:
        rc = _lib.lib().made_tape_replay(self.handle)
        if rc != 0 and self.callback_error is not None:
            ex, self.callback_error = self.callback_error, None
            raise ex
        _lib.check(rc, "made_tape_replay")

    def replay_range(self, first: int, count: int) -> None:
        """operations [first, first + count) only (one phase of the step on its own: measurements)"""
        _lib.check(_lib.lib().made_tape_replay_range(self.handle, int(first), int(count)), "made_tape_replay_range")

    def ops(self):
        """[(kind, function address, stream, (gx, gy, gz))] in issue order; kind 0 = kernel launch"""
        k, w, o = self.counts()
        out = []
        kind, fn, st, g3 = _lib.i32(0), C.c_uint64(0), C.c_uint64(0), (C.c_uint32 * 3)()
        for i in range(k + w + o):
            _lib.check(_lib.lib().made_tape_op(self.handle, i, C.byref(kind), C.byref(fn), C.byref(st), g3), "made_tape_op")
            out.append((int(kind.value), int(fn.value), int(st.value), (int(g3[0]), int(g3[1]), int(g3[2]))))
        return out

    def interleave(self, main_weight: int = 2) -> None:
        """feed all streams at the same time on replay (made_tape_interleave): results unchanged, only the host's issue order moves"""
        _lib.check(_lib.lib().made_tape_interleave(self.handle, int(main_weight)), "made_tape_interleave")

    def counts(self):
        k, w, o = _lib.i64(0), _lib.i64(0), _lib.i64(0)
        _lib.check(_lib.lib().made_tape_count(self.handle, C.byref(k), C.byref(w), C.byref(o)), "made_tape_count")
        return int(k.value), int(w.value), int(o.value)

    def close(self, _sync: bool = True) -> None:
        """Frees the tape and drops the tensors it kept alive -- after the device has finished: the last replay's kernels may still be
        queued on the tape's streams, and the caching allocator would hand the released memory to new allocations under them."""
        if self.handle.value:
            if _sync and torch.cuda.is_available() and torch.cuda.is_initialized():
                torch.cuda.synchronize()
            _lib.lib().made_tape_free(self.handle)
            self.handle = C.c_uint64(0)
        self._keep = []
        if _sync:
            _release_deferred()

    def __del__(self):
        # A finaliser can run at any point of the host program -- inside a hipGraph capture or another tape's recording, where a device
        # synchronize would invalidate the capture.  There nothing is freed: the handle and the tensors it keeps alive move to a module-level
        # list and are released, behind a synchronize, at the next safe point (the next record() entry / exit, an explicit close(), or
        # interpreter exit) -- the last replay's kernels may still be queued on the tape's streams.
        try:
            busy = _recording is not None or (torch.cuda.is_available() and torch.cuda.is_initialized() and torch.cuda.is_current_stream_capturing())
            if busy and self.handle.value:
                _deferred.append((self.handle, self._keep))
                self.handle, self._keep = C.c_uint64(0), []
            else:
                self.close(_sync=True)
        except Exception:
            pass

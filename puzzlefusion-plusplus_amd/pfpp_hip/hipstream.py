"""Cheap cross-stream ordering for the training engine's side streams.

`side.wait_stream(torch.cuda.current_stream()); with torch.cuda.stream(side): launch(...)` costs ~30 us of Python per use
(stream objects, a fresh Event, two context switches of torch's current stream) and the backward does it ~50 times per
iteration — a quarter of the host time of a step that is within 10 % of being host-bound.  Here the same ordering is two
direct HIP calls on a recycled event (hipEventRecord on the producing stream, hipStreamWaitEvent on the consuming one), and the
kernels are sent to the side stream through ops.stream_override (the wrappers' stream argument) without touching torch's
current stream.  Only for regions that launch pfpp kernels into existing buffers: torch ops and allocations still follow
torch's current stream."""
from __future__ import annotations

import ctypes as C
import os

import torch

_hip = None
_events = {}


def _lib():
    global _hip
    if _hip is None:
        bundled = os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so")
        try:                                          # the runtime torch itself runs on: the wheel's copy, else the one already loaded
            _hip = C.CDLL(bundled) if os.path.exists(bundled) else C.CDLL("libamdhip64.so")
        except OSError as e:
            raise RuntimeError("pfpp_hip.hipstream: cannot open the HIP runtime torch runs on (libamdhip64.so)") from e
        _hip.hipEventCreateWithFlags.argtypes = [C.POINTER(C.c_void_p), C.c_uint]
        _hip.hipEventRecord.argtypes = [C.c_void_p, C.c_void_p]
        _hip.hipStreamWaitEvent.argtypes = [C.c_void_p, C.c_void_p, C.c_uint]
    return _hip


def _event_ring(key, n: int = 64):
    """n recycled timing-free events per (producer, consumer) pair.  Re-recording an event does not disturb waits already
    enqueued on it (a wait captures the record that preceded it); the ring only keeps us far away from any doubt."""
    ring = _events.get(key)
    if ring is None:
        hip = _lib()
        evs = []
        for _ in range(n):
            e = C.c_void_p()
            if hip.hipEventCreateWithFlags(C.byref(e), 0x2) != 0:        # hipEventDisableTiming
                raise RuntimeError("hipEventCreateWithFlags failed")
            evs.append(e)
        ring = _events[key] = [evs, 0]
    evs, i = ring
    ring[1] = (i + 1) % len(evs)
    return evs[i]


def wait_for(consumer: int, producer: int) -> None:
    """everything queued on the raw stream `producer` so far happens before whatever is queued on `consumer` next"""
    hip = _lib()
    ev = _event_ring((producer, consumer))
    if hip.hipEventRecord(ev, C.c_void_p(producer)) != 0 or hip.hipStreamWaitEvent(C.c_void_p(consumer), ev, 0) != 0:
        raise RuntimeError("hipEventRecord / hipStreamWaitEvent failed")

"""PyTorch-ROCm host side of the hot path: autograd surface over the C ABI (ctypes, raw device pointers).

Mirrors the reference's operator interface (same names, argument meaning, defaults and error behaviour):

    render(pos, color, opacity_raw, sigma, c2w, H, W, fx, fy, cx, cy, near=0.01, far=100.0, pix_guard=32, T=16,
           min_conis=1e-6, chi_square_clip=6.25, alpha_max=0.99, alpha_cutoff=1/128.)      reference render.py:62-64
    build_sigma_from_params(scale_raw, q_raw)                                               reference gaussian.py:71
    evaluate_sh(f_dc, f_rest, points, c2w)                                   reference spherical_harmonics.py:70

plus the fused entry `render_gaussians(...)`, which takes the six raw parameter tensors and folds the covariance build
and the SH evaluation into the projection kernel (the Sigma[N,3,3] and colour[N,3] tensors are never materialised).

PyTorch is plumbing here (device memory, streams, autograd bookkeeping); all arithmetic runs in the HIP library.
There is no CPU path: CPU tensors, or a missing library, raise.
"""
import contextlib
import ctypes as C
import weakref

import numpy as np
import torch

from . import _abi

OFFSCREEN_MSG = "All projected points are off-screen"      # reference render.py:236


def _stream_ptr(device):
    return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _f32(t, shape, name):
    """Detached, contiguous fp32 view/copy of a tensor argument (reference tensors are already fp32 contiguous)."""
    if not isinstance(t, torch.Tensor):
        raise TypeError(f"{name} must be a torch.Tensor")
    # (the usual case first: an fp32, contiguous, aligned GPU tensor of the right shape is used as it stands -- only its
    #  address travels to the library, autograd never sees it)
    if t.dtype is torch.float32 and t.is_cuda and t.shape == shape and t.is_contiguous() and not t.data_ptr() % 16:
        return t
    if not t.is_cuda:
        raise RuntimeError(f"{name} is on {t.device}: the MI355X rasterizer needs GPU tensors (there is no CPU fallback)")
    if tuple(t.shape) != tuple(shape):
        raise ValueError(f"{name} has shape {tuple(t.shape)}, expected {tuple(shape)}")
    t = t.detach()
    if t.dtype != torch.float32:
        t = t.float()
    t = t.contiguous()
    if t.data_ptr() % 16:            # the kernels stage rows with 16-byte accesses (views into a larger storage may be offset)
        t = t.clone()
    return t


class PairCapacityExceeded(RuntimeError):
    """A frame rendered without waiting for its pair count (deferred_checks) had more (list, Gaussian) pairs than the buffers
    kept from earlier frames: its image and gradients are invalid.  The capacity has been raised; render it again."""


PINNED_SLOTS = 256      # counter blocks in flight per (device, stream) before one is reused


def capacity_key(device, view, n):
    """Pair capacities are kept per (device, image size, power-of-two bucket of the Gaussian count): one large scene does not
    make every later frame of a small one allocate, launch over and (deterministic mode) clear its buffers."""
    return (device.type, device.index, view.H, view.W, max(int(n), 1).bit_length())


class _Workspace:
    """Grow-only scratch buffers, the persistent counter block of gsplat_project, a ring of pinned counter blocks and one
    event per (device, stream): calls on different streams never share them (scratch is dead after each call, in stream
    order).  `capacity`: per capacity_key(), the largest pair count seen x 1.25 (sizes the buffers of frames that do not wait)."""

    def __init__(self):
        self.scratch = {}
        self.pinned = {}
        self.events = {}
        self.counters = {}
        self.capacity = {}
        self.slot = {}
        self.owners = {}            # per (device, stream): slot -> weak reference to the DeferredChecks that still has to read it
        self.event_pool = {}        # per device: events of frames whose counters have been read, handed out again (get_event(fresh=True))
        self.sizes = {}             # (n, capacity, H, W, flags) -> (frame bytes, bin scratch bytes): host arithmetic, cached

    def get_counter_block(self, device, nbytes, key=None):
        """Zeroed once; every gsplat_project call leaves it zeroed again (include/gsplat_mi355x.h).  (`key`: the caller's
        _key(device), when it has it already -- looking up the current stream is the costliest thing these methods do.)"""
        key = key or self._key(device)
        buf = self.counters.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.zeros(int(nbytes), dtype=torch.uint8, device=device)
            self.counters[key] = buf
        return buf

    def next_pinned(self, device, key=None, owner=None):
        """A pinned, device-mapped counter block nobody else is using.  A slot whose last frame still waits to be looked at by
        its DeferredChecks (a block left without verify(), or a very long one) is read by that owner first -- the frame
        finished long ago -- so a ring that comes round never hands out a block somebody still has to read."""
        key = key or self._key(device)
        ring = self.pinned.get(key)
        if ring is None:
            ring = self.pinned[key] = torch.zeros((PINNED_SLOTS, C.sizeof(_abi.Counts)), dtype=torch.uint8).pin_memory()
            self.owners[key] = {}
        i = self.slot.get(key, 0)
        self.slot[key] = (i + 1) % PINNED_SLOTS
        owners = self.owners[key]
        ref = owners.pop(i, None)
        if ref is not None:
            prev = ref()
            if prev is not None:
                prev._release_slot(key, i)
        if owner is not None:
            owners[i] = weakref.ref(owner)
        return ring[i], i

    def note_pairs(self, ckey, n_binned):
        want = int(n_binned * 1.25) + 4096
        if want > self.capacity.get(ckey, 0):
            self.capacity[ckey] = want

    def pair_capacity(self, ckey):
        return self.capacity.get(ckey, 0)

    def reset_pair_capacity(self):
        self.capacity.clear()

    @staticmethod
    def _key(device):
        return (device.type, device.index, torch.cuda.current_stream(device).cuda_stream)

    def get_event(self, device, fresh=False, key=None):
        """One reusable event per stream: gsplat_project records it right behind the counters (fresh: an event of its own,
        for a frame whose counters are read later)."""
        if fresh:
            pool = self.event_pool.get((device.type, device.index))
            if pool:
                return pool.pop()             # (its handle exists, and the frame it belonged to is long done)
        key = key or self._key(device)
        ev = None if fresh else self.events.get(key)
        if ev is None:
            ev = torch.cuda.Event(enable_timing=False, blocking=False)
            ev.record(torch.cuda.current_stream(device))          # events are created lazily: force the handle to exist
            if not fresh:
                self.events[key] = ev
        return ev

    def recycle_event(self, device, ev):
        pool = self.event_pool.setdefault((device.type, device.index), [])
        if len(pool) < 4 * PINNED_SLOTS:
            pool.append(ev)

    def get_scratch(self, device, nbytes, key=None):
        key = key or self._key(device)
        buf = self.scratch.get(key)
        if buf is None or buf.numel() < nbytes:
            buf = torch.empty(int(nbytes * 1.25) + 1024, dtype=torch.uint8, device=device)
            self.scratch[key] = buf
        return buf

    def frame_sizes(self, lib, n, capacity, view, flags):
        k = (n, capacity, view.H, view.W, flags)
        got = self.sizes.get(k)
        if got is None:
            if len(self.sizes) > 4096:
                self.sizes.clear()
            got = self.sizes[k] = (lib.gsplat_frame_bytes(n, capacity, C.byref(view), flags),
                                   lib.gsplat_bin_scratch_bytes(capacity, C.byref(view)))
        return got


def reset_pair_capacity():
    """Forget the pair capacities kept from earlier frames (a new scene, a new Trainer): the next frame of every (device, image
    size, Gaussian-count bucket) waits for its own count again."""
    _ws.reset_pair_capacity()


_ws = _Workspace()


class StageTimer:
    """Optional per-stage timing with HIP events recorded on the stream the kernels are launched on (torch's current
    stream).  bench.py installs one with `set_stage_timer`; when none is installed the hooks cost nothing."""

    def __init__(self, only=None, every=1):
        self.events = []          # (stage, start_event, end_event)
        self.only = set(only) if only else None      # restrict to these stages (every event pair costs ~10 us of stream time)
        self.every = max(1, int(every))              # ... and to every n-th pass (the passes in between run exactly as without a timer)
        self.passes = 0

    def wants(self, stages):
        """Does this pass bracket any of `stages`?  (asked ONCE per pass and direction by the render op; counts the passes)"""
        if self.only is not None and not (self.only & stages):
            return False
        self.passes += 1
        return (self.passes - 1) % self.every == 0

    def totals_ms(self):
        """stage -> (launches, total milliseconds); call after a device synchronise."""
        out = {}
        for name, a, b in self.events:
            n, t = out.get(name, (0, 0.0))
            out[name] = (n + 1, t + a.elapsed_time(b))
        return out

    def reset(self):
        self.events = []


_timer = None


def set_stage_timer(timer):
    global _timer
    _timer = timer


class _stage:
    __slots__ = ("name", "a")

    def __init__(self, name):
        self.name = name

    def __enter__(self):
        self.a = None
        if _timer is not None and (_timer.only is None or self.name in _timer.only):
            self.a = torch.cuda.Event(enable_timing=True)
            self.a.record()

    def __exit__(self, *exc):
        if self.a is not None and _timer is not None:
            b = torch.cuda.Event(enable_timing=True)
            b.record()
            _timer.events.append((self.name, self.a, b))
        return False


class DeferredChecks:
    """Render without waiting for the per-frame counters (training loops, frame sequences):

        with ops.deferred_checks() as chk:
            for view in views: loss(render_gaussians(...)).backward()      # no host synchronisation per view
        chk.verify()                                                        # ONE wait, then every frame's checks

    Inside the block a render sizes its pair buffers from the capacity kept from earlier frames (x 1.25 of the largest
    count seen for that image size and Gaussian-count bucket) instead of waiting for its own count, and the SH colour is
    evaluated inside the projection kernel (one pass over the inputs).  What the reference decides from the counts is decided
    in verify(): survivors but none on screen -> the same Exception("All projected points are off-screen"), now raised there; no
    survivor -> the frame was a zero image with zero gradients anyway.  A frame with more pairs than the capacity raises
    PairCapacityExceeded (capacity raised; render the block again: its results are invalid).  The first render of a size on a
    device (no capacity known yet) waits like an ordinary one.

    A block that is left by an exception has its frames' counters read then and there (nothing is raised on top of the exception
    in flight); one that is simply never verified keeps its findings, and its pinned counter blocks are read before the ring
    hands them to another frame (_Workspace.next_pinned): no block is ever reused unread."""

    def __init__(self):
        self.pending = []          # [pinned counter block, event, capacity, device, capacity key, (ring key, slot)]
        self.counts = []
        self._overflow = self._offscreen = False

    def add(self, pinned, event, capacity, device, ckey, slot):
        self.pending.append((pinned, event, capacity, device, ckey, slot))
        if len(self.pending) >= PINNED_SLOTS // 2:           # long sequences: look at the oldest frames before their pinned
            self._drain(len(self.pending) // 2)              # counter blocks come round again (they finished long ago)

    def _drain(self, count):
        global _last_counts, _last_binned
        for pinned, ev, cap, dev, ckey, slot in self.pending[:count]:
            ev.synchronize()
            counts = _abi.Counts.from_buffer_copy(pinned.numpy().tobytes())
            self.counts.append(counts)
            _ws.note_pairs(ckey, counts.n_binned)
            self._overflow |= cap is not None and counts.n_binned > cap
            self._offscreen |= _abi.lib().gsplat_classify_counts(C.byref(counts)) == _abi.GSPLAT_SCENE_ALL_OFFSCREEN
            _last_counts = (counts.n_survivors, counts.n_visible, int(counts.n_pairs))
            _last_binned = int(counts.n_binned)
            _ws.recycle_event(dev, ev)
            owners = _ws.owners.get(slot[0])
            if owners is not None:
                ref = owners.get(slot[1])
                if ref is not None and ref() is self:
                    del owners[slot[1]]
        del self.pending[:count]

    def _release_slot(self, ring_key, index):
        """The ring is about to hand slot `index` to another frame: read everything up to the frame that holds it."""
        for k, entry in enumerate(self.pending):
            if entry[5] == (ring_key, index):
                self._drain(k + 1)
                return

    def __enter__(self):
        _deferred_stack.append(self)
        return self

    def __exit__(self, exc_type, exc, tb):
        _deferred_stack.remove(self)
        if exc_type is not None and self.pending:           # nobody will call verify(): read the counters now, raise nothing
            try:
                self._drain(len(self.pending))
            except Exception:                                # (a device error while draining must not mask the exception in flight)
                self.pending = []
        return False

    def verify(self):
        self._drain(len(self.pending))
        overflow, offscreen = self._overflow, self._offscreen
        self._overflow = self._offscreen = False
        if offscreen:
            raise Exception(OFFSCREEN_MSG)
        if overflow:
            raise PairCapacityExceeded("more (list, Gaussian) pairs than the buffers kept from earlier frames: render the block again")
        return self.counts


_deferred_stack = []
forward_modes = {"waited": 0, "deferred": 0}     # forward passes that waited for their counters / that did not (diagnostics, tests)
composite_calls = {"forward": 0, "backward": 0}  # passes queued through ONE library call (gsplat_forward_deferred / gsplat_backward)


def deferred_checks():
    return DeferredChecks()


def run_deferred(fn, attempts=4):
    """fn() queues renders (and their backward passes) -- it runs inside deferred_checks() and is REPEATED while one of its frames
    outgrows the pair buffers kept from earlier frames (the capacity is raised each time; the first frame of a much larger scene
    does that once).  Returns fn()'s result of the pass that passed its checks; the off-screen exception propagates."""
    for _ in range(attempts):
        with deferred_checks() as chk:
            out = fn()
        try:
            chk.verify()
            return out
        except PairCapacityExceeded:
            continue
    raise RuntimeError(f"the pair buffers overflowed {attempts} times in a row")


def _make_gaussians(n, pos, opacity_raw, color=None, sigma=None, scale_raw=None, q_raw=None, f_dc=None, f_rest=None):
    return _abi.Gaussians(n, _p(pos), _p(opacity_raw), _p(color), _p(sigma), _p(scale_raw), _p(q_raw), _p(f_dc), _p(f_rest))


_sh_jacobian = True      # tests / ablations switch it off: the backward then reads the SH coefficients again (same gradients)
_composite = True        # tests / ablations switch it off: a deferred frame then goes through the separate library calls


class _Frame:
    """Everything the backward pass needs from one forward call.  A frame queued by gsplat_forward_deferred keeps ONE arena
    (project_state | bin_state | accum | grad2d, carved by the library); one that went through the separate calls keeps them
    as separate buffers."""
    __slots__ = ("view", "n", "n_pairs", "proj_state", "bin_state", "accum", "fused", "inputs", "c2w", "empty", "grad2d", "sh_jacobian",
                 "arena", "gaussians", "dirty", "src_ptrs")


class _Pending:
    """A forward call between its two halves: projection queued, counters not read yet."""
    __slots__ = ("frame", "gaussians", "pinned", "ready", "device", "stream", "capacity", "key", "st", "ckey", "slot")


def _convert_inputs(fused, n, pos, opacity_raw, c2w, a, b, c, d):
    pos32 = _f32(pos, (n, 3), "pos")
    opa32 = _f32(opacity_raw if opacity_raw.dim() == 1 else opacity_raw.reshape(-1), (n,), "opacity_raw")
    c2w32 = _f32(c2w, (4, 4), "c2w")
    if fused:
        ins = dict(scale_raw=_f32(a, (n, 3), "scale_raw"), q_raw=_f32(b, (n, 4), "q_raw"), f_dc=_f32(c, (n, 3), "f_dc"),
                   f_rest=_f32(d, (n, 45), "f_rest"))
    else:
        ins = dict(color=_f32(a, (n, 3), "color"), sigma=_f32(b, (n, 3, 3), "sigma"))
    return pos32, opa32, c2w32, ins


def _new_frame(fused, view, n, pos32, opa32, c2w32, ins, c, d):
    fr = _Frame()
    fr.view, fr.n, fr.fused, fr.c2w, fr.empty, fr.sh_jacobian = view, n, fused, c2w32, False, False
    fr.inputs = dict(pos=pos32, opacity_raw=opa32, **ins)
    fr.arena = fr.gaussians = fr.proj_state = fr.bin_state = fr.accum = fr.grad2d = None
    fr.dirty = False
    # the caller's own SH tensors (before any dtype / layout conversion): what dp.FactoredExchange.owns() compares
    fr.src_ptrs = (c.data_ptr(), d.data_ptr()) if fused else None
    return fr


def _forward_begin(fused, view, c2w, pos, opacity_raw, a, b, c, d, need_grad=False):
    """First half of the forward pass: everything up to (not including) the host's wait for the counters.  Returns
    (pending, None), or (None, result) when nothing is left to do: zero Gaussians, or -- inside a deferred_checks() block once a
    pair capacity is known for this image size -- the whole forward pass was queued by ONE library call
    (gsplat_forward_deferred) and the result is there."""
    lib = _abi.lib()
    dev = pos.device
    n = pos.shape[0]
    pos32, opa32, c2w32, ins = _convert_inputs(fused, n, pos, opacity_raw, c2w, a, b, c, d)
    fr = _new_frame(fused, view, n, pos32, opa32, c2w32, ins, c, d)
    if n == 0:      # nothing survives by construction: the reference returns the zero image (render.py:109-112)
        fr.empty = True
        return None, (torch.zeros((view.H, view.W, 3), dtype=torch.float32, device=dev), fr, _abi.Counts(0, 0, 0, 0, 0, 0))
    g = _make_gaussians(n, pos32, opa32, **ins)
    ckey = capacity_key(dev, view, n)
    capacity = _ws.pair_capacity(ckey) if _deferred_stack else 0
    deferred = capacity > 0
    if torch.cuda.current_device() != dev.index:
        torch.cuda.set_device(dev)            # (a context manager per call costs more than the call: the one-process-per-GPU host never switches)
    stream = torch.cuda.current_stream(dev)                       # looked up ONCE per forward pass
    sp = stream.cuda_stream
    key = (dev.type, dev.index, sp)
    st = C.c_void_p(sp)
    counters = _ws.get_counter_block(dev, _COUNTER_BYTES or _counter_bytes(lib), key)
    fr.sh_jacobian = bool(fused and need_grad and _sh_jacobian)   # 48 bytes per Gaussian that spare the backward the 192 bytes of SH coefficients
    chk = _deferred_stack[-1] if deferred else None
    pinned, slot = _ws.next_pinned(dev, key, chk)
    ready = _ws.get_event(dev, fresh=deferred, key=key)
    wants_stages = _timer is not None and _timer.wants(_FORWARD_STAGES)
    if deferred and _composite and not wants_stages:
        # ---- the whole forward pass in one call, on one arena
        H, W = view.H, view.W
        flags = (_abi.GSPLAT_FRAME_BACKWARD if need_grad else 0) | (0 if _sh_jacobian else _abi.GSPLAT_FRAME_NO_SH_JACOBIAN)
        frame_bytes, scratch_bytes = _ws.frame_sizes(lib, n, capacity, view, flags)
        fr.arena = torch.empty(frame_bytes, dtype=torch.uint8, device=dev)
        image = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
        scratch = _ws.get_scratch(dev, scratch_bytes, key)
        fr.gaussians, fr.n_pairs = g, capacity
        _abi.check(lib.gsplat_forward_deferred(g, c2w32.data_ptr(), view, fr.arena.data_ptr(), frame_bytes, capacity, counters.data_ptr(),
                                               counters.numel(), scratch.data_ptr(), scratch.numel(), pinned.data_ptr(), ready.cuda_event,
                                               image.data_ptr(), flags, st), "gsplat_forward_deferred")
        forward_modes["deferred"] += 1
        composite_calls["forward"] += 1
        chk.add(pinned, ready, capacity, dev, ckey, (key, slot))
        return None, (image, fr, None)
    pend = _Pending()
    pend.frame, pend.gaussians, pend.device, pend.stream, pend.key, pend.st = fr, g, dev, stream, key, st
    pend.capacity = capacity if deferred else None
    pend.pinned, pend.ready, pend.ckey, pend.slot = pinned, ready, ckey, (key, slot)
    fr.proj_state = torch.empty(lib.gsplat_project_state_bytes(n, C.byref(view)), dtype=torch.uint8, device=dev)
    # the counters go straight into the pinned block (mapped into the device's address space: no copy operation);
    # a frame that will not wait for them evaluates the SH colour inside the projection kernel and lets the first binning
    # kernel total the counters (the projection's waves then retire without waiting for their stores)
    flags = _abi.GSPLAT_PROJECT_COUNTS_MAPPED | ((_abi.GSPLAT_PROJECT_COLOUR_FUSED | _abi.GSPLAT_PROJECT_COUNTS_LATE) if deferred else 0)
    if fr.sh_jacobian:
        flags |= _abi.GSPLAT_PROJECT_SAVE_SH_JACOBIAN
    with _stage("project"):
        _abi.check(lib.gsplat_project(C.byref(g), _p(c2w32), C.byref(view), _p(fr.proj_state), _p(counters),
                                      counters.numel(), C.c_void_p(pinned.data_ptr()), C.c_void_p(ready.cuda_event),
                                      flags, st), "gsplat_project")
    return pend, None


_COUNTER_BYTES = 0
_FORWARD_STAGES = frozenset(("project", "bin", "raster_forward"))


def _counter_bytes(lib):
    global _COUNTER_BYTES
    _COUNTER_BYTES = int(lib.gsplat_project_scratch_bytes(0))
    return _COUNTER_BYTES


def _forward_end(pend, need_grad):
    """Second half: wait for the counters, size the pair buffers, bin, rasterise.  Must run with the same current stream
    as the first half."""
    lib = _abi.lib()
    fr, dev = pend.frame, pend.device
    view, n = fr.view, fr.n
    H, W = view.H, view.W
    st = pend.st                                 # (the same current stream as in the first half: the caller's contract)
    # the one host wait of the forward pass: the pair count sizes the binning buffers, and the reference's empty /
    # off-screen conventions need the survivor counts.  Only the counters are waited for: the first binning kernel and
    # (fused inputs) the SH colour pass are queued behind them and run during this round trip.
    image = torch.empty((H, W, 3), dtype=torch.float32, device=dev)
    if pend.capacity is not None:
        # deferred: no wait.  Buffers of the capacity kept from earlier frames; the kernels read the real count on the
        # device; the host looks at the counters in DeferredChecks.verify()
        counts = None
        fr.n_pairs = int(pend.capacity)
        forward_modes["deferred"] += 1
        _deferred_stack[-1].add(pend.pinned, pend.ready, fr.n_pairs, dev, pend.ckey, pend.slot)
    else:
        pend.ready.synchronize()
        forward_modes["waited"] += 1
        counts = _abi.Counts.from_buffer_copy(pend.pinned.numpy().tobytes())
        _ws.note_pairs(pend.ckey, counts.n_binned)
        if _deferred_stack:                      # a frame that had to wait inside a deferred block (first of its size): verify() still
            chk = _deferred_stack[-1]            # returns one entry per frame, in frame order (the earlier frames are done by now)
            chk._drain(len(chk.pending))
            chk.counts.append(counts)
        scene = lib.gsplat_classify_counts(C.byref(counts))
        if scene == _abi.GSPLAT_SCENE_ALL_OFFSCREEN:
            raise Exception(OFFSCREEN_MSG)
        if scene == _abi.GSPLAT_SCENE_ALL_CULLED:
            fr.empty = True
            fr.proj_state = None
            return image.zero_(), fr, counts
        fr.n_pairs = int(counts.n_binned)        # pairs actually binned (16 x 8 lists); counts.n_pairs = the reference's P
    fr.bin_state = torch.empty(lib.gsplat_bin_state_bytes(fr.n_pairs, C.byref(view)), dtype=torch.uint8, device=dev)
    scratch = _ws.get_scratch(dev, lib.gsplat_bin_scratch_bytes(fr.n_pairs, C.byref(view)), pend.key)
    with _stage("bin"):
        _abi.check(lib.gsplat_bin(n, fr.n_pairs, C.byref(view), _p(fr.proj_state), _p(fr.bin_state), _p(scratch),
                                  scratch.numel(), st), "gsplat_bin")
    fr.accum = torch.empty((H, W, 3), dtype=torch.float32, device=dev) if need_grad else None
    # the forward rasterizer clears the backward's accumulation buffer on the side (its waves are VALU-bound), unless
    # there are so few lists that a wave's share would be long
    lists = ((W + 15) // 16) * ((H + 7) // 8)
    fr.grad2d = torch.empty((n, 16), dtype=torch.float32, device=dev) if need_grad and n <= 256 * lists else None
    with _stage("raster_forward"):
        _abi.check(lib.gsplat_rasterize_forward(n, fr.n_pairs, C.byref(view), _p(fr.proj_state), _p(fr.bin_state),
                                                _p(image), _p(fr.accum), _p(fr.grad2d), st), "gsplat_rasterize_forward")
    return image, fr, counts


def _forward_impl(fused, view, c2w, pos, opacity_raw, a, b, c, d, need_grad):
    pend, done = _forward_begin(fused, view, c2w, pos, opacity_raw, a, b, c, d, need_grad)
    return done if pend is None else _forward_end(pend, need_grad)


def _flat_like(ins):
    """One flat fp32 buffer holding a gradient for every input (each view 256-byte aligned inside it).  The data-parallel
    helper recognises the shared base and all-reduces the six gradients with ONE in-place collective, no flatten copy."""
    sizes = [(v.numel() + 63) // 64 * 64 for v in ins.values()]
    any_in = next(iter(ins.values()))
    flat = torch.empty(sum(sizes), dtype=torch.float32, device=any_in.device)
    parts = flat.split(sizes)                                   # one call for all the pieces
    return {k: (piece if piece.numel() == v.numel() else piece[:v.numel()]).view(v.shape) for (k, v), piece in zip(ins.items(), parts)}


# Data-parallel exchange of the SH gradients in factored form (dp.FactoredExchange): while a sink is installed
# the render backward of the fused entry hands the 3 colour-logit gradients per Gaussian to the sink instead of
# computing the 48 SH-coefficient gradients, and returns no gradient for f_dc / f_rest.
_sh_sink = None
_deterministic = False


def set_deterministic(flag=True):
    """Bitwise reproducible gradients: the raster backward stores its per-(list, Gaussian) sums and adds them per Gaussian in a
    fixed order instead of using float atomics (whose order of arrival changes the last bits from run to run).  Slower by the
    cost of a 36-byte row per pair written and read once.  Returns the previous setting."""
    global _deterministic
    old, _deterministic = _deterministic, bool(flag)
    return old


_rest_update = None
_grad_acc = None


class GradAccumulation:
    """with ops.accumulate_grads(params) as acc: ... several render(...).backward() ...; acc.assign()

    The gradients of the views of ONE iteration are summed by the projection backward itself (GSPLAT_BACKWARD_ACCUMULATE) in one flat
    buffer: the first view writes, the others add -- instead of autograd's AccumulateGrad pass per view (read two, write one: 0.4 ms
    per view at 3 M Gaussians).  `params`: dict name -> the leaf tensors the renders are called with (fp32, contiguous, .grad None).
    A render of other tensors, a frame that waited for its counters or a data-parallel sink keep the ordinary backward.  assign()
    sets param.grad (views of the buffer) -- summing in what the ordinary backward may have produced for some views."""

    NAMES = ("pos", "opacity_raw", "scale_raw", "q_raw", "f_dc", "f_rest")

    def __init__(self, params):
        self.params = {k: params[k] for k in self.NAMES}
        self.buffers = None
        self.count = 0
        self.event = None
        self.ptrs = {k: v.data_ptr() for k, v in self.params.items()}

    def matches(self, ins):
        return all(k in ins and ins[k].data_ptr() == self.ptrs[k] for k in self.NAMES)

    def begin(self, stream):
        """(buffers, accumulate?) for the next backward pass, ordered behind the previous one whatever stream that ran on."""
        if self.buffers is None:
            self.buffers = _flat_like({k: self.params[k] for k in self.NAMES})
        if self.event is not None:
            stream.wait_event(self.event)
        return self.buffers, self.count > 0

    def done(self, stream):
        self.count += 1
        if self.event is None:
            self.event = torch.cuda.Event(enable_timing=False)
        self.event.record(stream)

    def assign(self):
        if self.count == 0:
            return
        stream = torch.cuda.current_stream(self.params["pos"].device)
        stream.wait_event(self.event)
        for k, p in self.params.items():
            g = self.buffers[k]
            g.record_stream(stream)
            p.grad = g if p.grad is None else p.grad.add_(g)


@contextlib.contextmanager
def accumulate_grads(params):
    global _grad_acc
    acc = GradAccumulation(params)
    _grad_acc = acc
    try:
        yield acc
    finally:
        _grad_acc = None


def set_rest_update(hook):
    """hook (optim.GaussianAdam.fused_rest_update) or None: while one is installed, the backward pass of a deferred frame rendered
    from the hook's own f_rest tensor applies the Adam step of f_rest inside the projection backward (gsplat_backward_adam_rest) and
    returns no gradient for it."""
    global _rest_update
    _rest_update = hook


def set_sh_gradient_sink(sink):
    """sink: object with .add(grad_logit[N,3], eye[3] device tensor), or None to restore the ordinary backward."""
    global _sh_sink
    _sh_sink = sink


def _backward_impl(fr, grad_image):
    """Returns a dict name -> fp32 gradient tensor for every input of the forward call."""
    lib = _abi.lib()
    ins = fr.inputs
    dev = ins["pos"].device
    # the sink only takes over for the parameter tensors it was built for: a render of other tensors (an evaluation
    # model, a test) while a sink is installed keeps its ordinary SH gradients
    factored = fr.fused and _sh_sink is not None and getattr(_sh_sink, "owns", lambda _ins, _ptrs=None: True)(ins, fr.src_ptrs)
    if fr.empty or fr.n == 0:
        if factored:
            _sh_sink.add(torch.zeros((fr.n, 3), dtype=torch.float32, device=dev), fr.c2w[:3, 3])
            return {k: (None if k in ("f_dc", "f_rest") else torch.zeros_like(v)) for k, v in ins.items()}
        return {k: torch.zeros_like(v) for k, v in ins.items()}
    gi = _f32(grad_image, (fr.view.H, fr.view.W, 3), "grad_image")
    if torch.cuda.current_device() != dev.index:
        torch.cuda.set_device(dev)
    stream = torch.cuda.current_stream(dev)
    st = C.c_void_p(stream.cuda_stream)
    det = None
    if _deterministic:
        det = _ws.get_scratch(dev, lib.gsplat_rasterize_backward_scratch_bytes(fr.n, fr.n_pairs), (dev.type, dev.index, stream.cuda_stream))
    wants_stages = _timer is not None and _timer.wants(_BACKWARD_STAGES)
    upd = _rest_update
    fold_rest = (upd is not None and fr.arena is not None and not factored and not wants_stages and fr.fused and fr.sh_jacobian
                 and not upd.applied and fr.src_ptrs is not None and upd.matches(ins["f_rest"], fr.src_ptrs[1]))
    acc = _grad_acc
    sum_views = (acc is not None and fr.arena is not None and not factored and not wants_stages and not fold_rest and fr.fused
                 and fr.sh_jacobian and not fr.dirty and acc.matches(ins))
    if sum_views:
        # the views of an iteration summed in the accumulation's own buffer by the projection backward: autograd gets no gradient
        bufs, add = acc.begin(stream)
        gg = _abi.GaussianGrads(_p(bufs["pos"]), _p(bufs["opacity_raw"]), None, None, _p(bufs["scale_raw"]), _p(bufs["q_raw"]),
                                _p(bufs["f_dc"]), _p(bufs["f_rest"]))
        fr.dirty = True
        flags = _abi.GSPLAT_BACKWARD_SH_JACOBIAN | (_abi.GSPLAT_BACKWARD_ACCUMULATE if add else 0)
        _abi.check(lib.gsplat_backward(fr.gaussians, fr.c2w.data_ptr(), fr.view, fr.arena.data_ptr(), fr.arena.numel(), fr.n_pairs, gi.data_ptr(), gg,
                                       None, det.data_ptr() if det is not None else None, det.numel() if det is not None else 0, flags, st),
                   "gsplat_backward")
        acc.done(stream)
        composite_calls["backward"] += 1
        return {k: None for k in ins}
    out = _flat_like({k: v for k, v in ins.items() if not ((factored and k in ("f_dc", "f_rest")) or (fold_rest and k == "f_rest"))})
    gg = _abi.GaussianGrads(_p(out["pos"]), _p(out["opacity_raw"]), _p(None if factored else out.get("color")), _p(out.get("sigma")),
                            _p(out.get("scale_raw")), _p(out.get("q_raw")), _p(out.get("f_dc")), _p(out.get("f_rest")))
    jac = _abi.GSPLAT_BACKWARD_SH_JACOBIAN if fr.sh_jacobian else 0
    if fr.arena is not None:
        # ---- the frame was queued by gsplat_forward_deferred: one call for the whole backward pass (two with a sink in between)
        dirty = _abi.GSPLAT_BACKWARD_GRAD2D_DIRTY if fr.dirty else 0
        fr.dirty = True                        # a second backward through the same graph must not reuse a dirty buffer
        args = (fr.gaussians, fr.c2w.data_ptr(), fr.view, fr.arena.data_ptr(), fr.arena.numel(), fr.n_pairs, gi.data_ptr(), gg)
        dargs = (det.data_ptr() if det is not None else None, det.numel() if det is not None else 0)
        if fold_rest:
            # the Adam step of f_rest inside the projection backward: its 192 bytes of gradient per Gaussian are never written
            group, b1, b2, eps = upd.begin()
            _abi.check(lib.gsplat_backward_adam_rest(*args, *dargs, jac | dirty, C.byref(group), b1, b2, eps, st), "gsplat_backward_adam_rest")
            composite_calls["backward"] += 1
            out["f_rest"] = None
            return out
        if factored or wants_stages:
            glogit = torch.empty((fr.n, 3), dtype=torch.float32, device=dev) if factored else None
            with _stage("raster_backward"):
                _abi.check(lib.gsplat_backward(*args, _p(glogit), *dargs, jac | dirty | _abi.GSPLAT_BACKWARD_PHASE_RASTER, st), "gsplat_backward")
            if factored:       # logit gradients first: the sink may start exchanging them while the projection backward runs
                _sh_sink.add(glogit, fr.c2w[:3, 3])
            with _stage("project_backward"):
                _abi.check(lib.gsplat_backward(*args, None, None, 0, jac | _abi.GSPLAT_BACKWARD_PHASE_PROJECT, st), "gsplat_backward")
        else:
            _abi.check(lib.gsplat_backward(*args, None, *dargs, jac | dirty, st), "gsplat_backward")
        composite_calls["backward"] += 1
    else:
        zeroed = fr.grad2d is not None
        grad2d = fr.grad2d if zeroed else torch.empty((fr.n, 16), dtype=torch.float32, device=dev)
        fr.grad2d = None                       # a second backward through the same graph must not reuse a dirty buffer
        with _stage("raster_backward"):
            _abi.check(lib.gsplat_rasterize_backward(fr.n, fr.n_pairs, C.byref(fr.view), _p(fr.proj_state), _p(fr.bin_state),
                                                     _p(fr.accum), _p(gi), _p(grad2d), int(zeroed), _p(det),
                                                     det.numel() if det is not None else 0, st), "gsplat_rasterize_backward")
        if factored:
            # logit gradients first: the sink may start exchanging them while the projection backward runs
            glogit = torch.empty((fr.n, 3), dtype=torch.float32, device=dev)
            _abi.check(lib.gsplat_logit_grad(fr.n, C.byref(fr.view), _p(fr.proj_state), _p(grad2d), _p(glogit), st), "gsplat_logit_grad")
            _sh_sink.add(glogit, fr.c2w[:3, 3])
        g = _make_gaussians(fr.n, **ins)
        with _stage("project_backward"):
            _abi.check(lib.gsplat_project_backward(C.byref(g), _p(fr.c2w), C.byref(fr.view), _p(fr.proj_state), _p(grad2d),
                                                   C.byref(gg), jac, st), "gsplat_project_backward")
    if factored:
        out["f_dc"] = out["f_rest"] = None
    return out


_BACKWARD_STAGES = frozenset(("raster_backward", "project_backward"))


def sh_accumulate(pos, eyes, grad_logit, scale=1.0):
    """(grad_f_dc [N,3], grad_f_rest [N,45]) = scale * sum over views of grad_logit[v] (x) Y(direction from eyes[v] to pos)."""
    lib = _abi.lib()
    n = pos.shape[0]
    v = grad_logit.shape[0]
    pos32, eyes32, gl32 = _f32(pos, (n, 3), "pos"), _f32(eyes, (v, 3), "eyes"), _f32(grad_logit, (v, n, 3), "grad_logit")
    dev = pos32.device
    with torch.cuda.device(dev):
        off = (n * 3 + 63) // 64 * 64                     # both views 256-byte aligned inside one buffer
        flat = torch.empty(off + n * 45, dtype=torch.float32, device=dev)
        g_dc, g_rest = flat[:n * 3].view(n, 3), flat[off:off + n * 45].view(n, 45)
        _abi.check(lib.gsplat_sh_accumulate(n, v, _p(pos32), _p(eyes32), _p(gl32), float(scale), _p(g_dc), _p(g_rest),
                                            _stream_ptr(dev)), "gsplat_sh_accumulate")
    return g_dc, g_rest


class _RenderFn(torch.autograd.Function):
    """Autograd node for both entry points; gradients for the tensor inputs only (c2w and scalars get None)."""

    @staticmethod
    def forward(ctx, fused, view, c2w, pos, opacity_raw, a, b, c, d):
        # needs_input_grad ignores the grad mode (and forward() itself always runs with grad disabled): the caller's grad mode
        # travels in view.grad_mode.  Under torch.no_grad() nothing is saved for a backward that cannot come.
        need = view.grad_mode and any(ctx.needs_input_grad)
        image, fr, counts = _forward_impl(fused, view, c2w, pos, opacity_raw, a, b, c, d, need)
        ctx.frame = fr
        ctx.dtypes = [t.dtype if isinstance(t, torch.Tensor) else None for t in (pos, opacity_raw, a, b, c, d)]
        ctx.opa_shape = opacity_raw.shape
        global _last_counts, _last_binned
        if counts is not None:                   # (a deferred frame's counters are read in DeferredChecks.verify())
            ctx.counts = _last_counts = (counts.n_survivors, counts.n_visible, int(counts.n_pairs))
            _last_binned = int(counts.n_binned)
        return image if pos.dtype == torch.float32 else image.to(pos.dtype)

    @staticmethod
    def backward(ctx, grad_image):
        fr = ctx.frame
        g = _backward_impl(fr, grad_image)
        names = ("pos", "opacity_raw") + (("scale_raw", "q_raw", "f_dc", "f_rest") if fr.fused else ("color", "sigma", None, None))
        outs = []
        for i, nm in enumerate(names):
            if nm is None or not ctx.needs_input_grad[3 + i]:
                outs.append(None)
                continue
            t = g[nm]
            if t is None:                     # f_dc / f_rest while a factored-exchange sink is installed
                outs.append(None)
                continue
            if nm == "opacity_raw":
                t = t.reshape(ctx.opa_shape)
            outs.append(t if ctx.dtypes[i] == torch.float32 else t.to(ctx.dtypes[i]))
        return (None, None, None, *outs)


def _view(H, W, fx, fy, cx, cy, near, far, pix_guard, T, min_conis, chi_square_clip, alpha_max, alpha_cutoff):
    # H, W may arrive as 0-d tensors from DataLoader collate (reference scripts/train.py:499)
    # every T of the reference is accepted: the image does not depend on it; T only sets the reference's
    # tile rectangles, i.e. the reported pair count P.  The kernels always bin 16 x 8-pixel lists.
    if int(T) < 1:
        raise ValueError("tile size T must be >= 1")
    view = _abi.make_view(int(H), int(W), float(fx), float(fy), float(cx), float(cy), near, far, pix_guard, T, min_conis,
                          chi_square_clip, alpha_max, alpha_cutoff)
    view.grad_mode = torch.is_grad_enabled()            # Python-side attribute (not part of the C struct)
    return view


def render(pos, color, opacity_raw, sigma, c2w, H, W, fx, fy, cx, cy, near=0.01, far=100.0, pix_guard=32, T=16,
           min_conis=1e-6, chi_square_clip=6.25, alpha_max=0.99, alpha_cutoff=1 / 128.):
    """Drop-in for the reference render() (gaussian_splatting/render.py:62-410).

    Returns the image [H, W, 3] in [0, 1], same dtype/device as `pos`, differentiable w.r.t. pos, color, opacity_raw
    and sigma.  No opacity / frustum / finite survivor -> zero image with zero gradients; survivors but none on
    screen -> Exception("All projected points are off-screen"), as in the reference.
    """
    view = _view(H, W, fx, fy, cx, cy, near, far, pix_guard, T, min_conis, chi_square_clip, alpha_max, alpha_cutoff)
    return _RenderFn.apply(False, view, c2w, pos, opacity_raw, color, sigma, None, None)


def render_gaussians(pos, f_dc, f_rest, opacity_raw, scale_raw, q_raw, c2w, H, W, fx, fy, cx, cy, near=0.01, far=100.0,
                     pix_guard=32, T=16, min_conis=1e-6, chi_square_clip=6.25, alpha_max=0.99, alpha_cutoff=1 / 128.):
    """Fused entry: render(pos, evaluate_sh(f_dc, f_rest, pos, c2w), opacity_raw, build_sigma_from_params(scale_raw,
    q_raw), c2w, ...) in one pass (the reference's three-call sequence, scripts/train.py:463,502,505-508)."""
    view = _view(H, W, fx, fy, cx, cy, near, far, pix_guard, T, min_conis, chi_square_clip, alpha_max, alpha_cutoff)
    return _RenderFn.apply(True, view, c2w, pos, opacity_raw, scale_raw, q_raw, f_dc, f_rest)


@torch.no_grad()
def render_frames(pos, f_dc, f_rest, opacity_raw, scale_raw, q_raw, c2ws, H, W, fx, fy, cx, cy, near=0.01, far=100.0,
                  pix_guard=32, T=16, min_conis=1e-6, chi_square_clip=6.25, alpha_max=0.99, alpha_cutoff=1 / 128., on_frame=None):
    """Forward-only rendering of a sequence of camera poses with the frames software-pipelined over two HIP streams:
    frame k + 1's projection / binning front (latency- and bandwidth-bound) overlaps frame k's rasterisation (VALU-bound).
    Same images as render_gaussians() frame by frame.  Returns the list of images (or calls on_frame(k, image) and returns
    None); the caller's current stream waits for all of them.  Without on_frame, and once a pair capacity is known for the
    device, no frame waits for its counters either (deferred_checks: the per-frame checks are made after the last frame is
    queued; a sequence that outgrows the buffers is rendered again)."""
    view = _view(H, W, fx, fy, cx, cy, near, far, pix_guard, T, min_conis, chi_square_clip, alpha_max, alpha_cutoff)
    dev = pos.device
    cams = [torch.as_tensor(c, dtype=torch.float32, device=dev) if not isinstance(c, torch.Tensor) else c for c in c2ws]
    args = (view, dev, cams, pos, f_dc, f_rest, opacity_raw, scale_raw, q_raw)
    if on_frame is not None or _ws.pair_capacity(capacity_key(dev, view, pos.shape[0])) == 0 or _deferred_stack:
        return _render_frames(*args, on_frame)
    return run_deferred(lambda: _render_frames(*args, None))


def _render_frames(view, dev, cams, pos, f_dc, f_rest, opacity_raw, scale_raw, q_raw, on_frame):
    main = torch.cuda.current_stream(dev)
    streams = _pipeline_streams(dev)
    for st in streams:
        st.wait_stream(main)                                   # parameters and camera matrices produced on the caller's stream

    def begin(k):
        with torch.cuda.stream(streams[k % 2]):
            return _forward_begin(True, view, cams[k], pos, opacity_raw, scale_raw, q_raw, f_dc, f_rest)

    def end(k, started):
        pend, done = started
        with torch.cuda.stream(streams[k % 2]):
            image = (done if pend is None else _forward_end(pend, False))[0]
        return image

    images = []
    started = begin(0) if cams else None
    for k in range(len(cams)):
        nxt = begin(k + 1) if k + 1 < len(cams) else None      # queue the next frame's front before waiting for this one's counters
        image = end(k, started)
        started = nxt
        image.record_stream(main)                              # allocated on a side stream, consumed on the caller's
        if on_frame is not None:
            main.wait_stream(streams[k % 2])                   # GPU-side dependency only: the host does not block
            on_frame(k, image)
        else:
            images.append(image)
    for st in streams:
        main.wait_stream(st)
    return None if on_frame is not None else images


_pipe_streams = {}


def _pipeline_streams(dev):
    key = (dev.type, dev.index)
    got = _pipe_streams.get(key)
    if got is None:
        got = _pipe_streams[key] = (torch.cuda.Stream(dev), torch.cuda.Stream(dev))
    return got


_last_counts = None
_last_binned = None


def binned_pairs():
    """(list, Gaussian) pairs the most recent call really binned (after the exact ellipse / list test): P_b of DESIGN.md."""
    return _last_binned


def render_stats(image=None):
    """(n_survivors, n_visible V, n_pairs P) of the call that produced `image` (or of the most recent call)."""
    fn = image.grad_fn if image is not None else None
    got = getattr(fn, "counts", None) if fn is not None else None
    return got if got is not None else _last_counts


class _BuildSigmaFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, scale_raw, q_raw):
        lib = _abi.lib()
        n = scale_raw.shape[0]
        sr, qr = _f32(scale_raw, (n, 3), "scale_raw"), _f32(q_raw, (n, 4), "q_raw")
        out = torch.empty((n, 3, 3), dtype=torch.float32, device=sr.device)
        with torch.cuda.device(sr.device):
            _abi.check(lib.gsplat_build_sigma(n, _p(sr), _p(qr), _p(out), _stream_ptr(sr.device)), "gsplat_build_sigma")
        ctx.save_for_backward(sr, qr)
        ctx.dtypes = (scale_raw.dtype, q_raw.dtype)
        return out if scale_raw.dtype == torch.float32 else out.to(scale_raw.dtype)

    @staticmethod
    def backward(ctx, grad_sigma):
        lib = _abi.lib()
        sr, qr = ctx.saved_tensors
        n = sr.shape[0]
        gs = _f32(grad_sigma, (n, 3, 3), "grad_sigma")
        gsr, gqr = torch.empty_like(sr), torch.empty_like(qr)
        with torch.cuda.device(sr.device):
            _abi.check(lib.gsplat_build_sigma_backward(n, _p(sr), _p(qr), _p(gs), _p(gsr), _p(gqr), _stream_ptr(sr.device)),
                       "gsplat_build_sigma_backward")
        return gsr.to(ctx.dtypes[0]), gqr.to(ctx.dtypes[1])


def build_sigma_from_params(scale_raw, q_raw):
    """Drop-in for the reference build_sigma_from_params (gaussian_splatting/gaussian.py:71-127): Sigma = R S S R^T."""
    return _BuildSigmaFn.apply(scale_raw, q_raw)


class _EvaluateShFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, f_dc, f_rest, points, c2w):
        lib = _abi.lib()
        n = points.shape[0]
        if f_rest.shape[-1] != 45:
            # the reference raises RuntimeError for any other width
            raise RuntimeError(f"evaluate_sh needs f_rest of width 45 (degree-3 SH), got {tuple(f_rest.shape)}")
        dc, rest = _f32(f_dc, (n, 3), "f_dc"), _f32(f_rest, (n, 45), "f_rest")
        pts, cam = _f32(points, (n, 3), "points"), _f32(c2w, (4, 4), "c2w")
        out = torch.empty((n, 3), dtype=torch.float32, device=pts.device)
        with torch.cuda.device(pts.device):
            _abi.check(lib.gsplat_evaluate_sh(n, _p(dc), _p(rest), _p(pts), _p(cam), _p(out), _stream_ptr(pts.device)),
                       "gsplat_evaluate_sh")
        ctx.save_for_backward(dc, rest, pts, cam)
        ctx.dtypes = (f_dc.dtype, f_rest.dtype, points.dtype)
        return out if points.dtype == torch.float32 else out.to(points.dtype)

    @staticmethod
    def backward(ctx, grad_color):
        lib = _abi.lib()
        dc, rest, pts, cam = ctx.saved_tensors
        n = pts.shape[0]
        gc = _f32(grad_color, (n, 3), "grad_color")
        gdc, grest, gpts = torch.empty_like(dc), torch.empty_like(rest), torch.empty_like(pts)
        with torch.cuda.device(pts.device):
            _abi.check(lib.gsplat_evaluate_sh_backward(n, _p(dc), _p(rest), _p(pts), _p(cam), _p(gc), _p(gdc), _p(grest),
                                                       _p(gpts), _stream_ptr(pts.device)), "gsplat_evaluate_sh_backward")
        return gdc.to(ctx.dtypes[0]), grest.to(ctx.dtypes[1]), gpts.to(ctx.dtypes[2]), None


def evaluate_sh(f_dc, f_rest, points, c2w):
    """Drop-in for the reference evaluate_sh (gaussian_splatting/spherical_harmonics.py:70-166): degree-3 real SH,
    channel-major f_rest, sigmoid output."""
    return _EvaluateShFn.apply(f_dc, f_rest, points, c2w)


# ---- the small helper functions the reference namespace also exports (host-side, not on the hot path) ----

HARMONICS = {   # reference gaussian_splatting/spherical_harmonics.py:50-67
    'SH_C0': 0.28209479177387814, 'SH_C1_x': 0.4886025119029199, 'SH_C1_y': 0.4886025119029199,
    'SH_C1_z': 0.4886025119029199, 'SH_C2_xy': 1.0925484305920792, 'SH_C2_xz': 1.0925484305920792,
    'SH_C2_yz': 1.0925484305920792, 'SH_C2_zz': 0.31539156525252005, 'SH_C2_xx_yy': 0.5462742152960396,
    'SH_C3_yxx_yyy': 0.5900435899266435, 'SH_C3_xyz': 2.890611442640554, 'SH_C3_yzz_yxx_yyy': 0.4570457994644658,
    'SH_C3_zzz_zxx_zyy': 0.3731763325901154, 'SH_C3_xzz_xxx_xyy': 0.4570457994644658,
    'SH_C3_zxx_zyy': 1.445305721320277, 'SH_C3_xxx_xyy': 0.5900435899266435,
}


def quat_to_rotmat(quat):
    """(x, y, z, w) quaternions [..., 4] -> rotation matrices [..., 3, 3], no normalisation (reference gaussian.py:24-68)."""
    x, y, z, w = quat.unbind(-1)
    m = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w),
                     2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w),
                     2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)], -1)
    return m.reshape(quat.shape[:-1] + (3, 3))


def inv2x2(M, eps=1e-12):
    """Batched 2x2 inverse with the determinant clamped at eps (reference utils.py:152-191)."""
    det = (M[:, 0, 0] * M[:, 1, 1] - M[:, 0, 1] * M[:, 1, 0]).clamp(min=eps)
    adj = torch.stack([M[:, 1, 1], -M[:, 0, 1], -M[:, 1, 0], M[:, 0, 0]], -1).reshape(-1, 2, 2)
    return adj / det.reshape(-1, 1, 1)


def scale_intrinsics(H, W, H_src, W_src, fx, fy, cx, cy):
    """Rescale pinhole intrinsics to another resolution (reference utils.py:194-238)."""
    sx, sy = W / W_src, H / H_src
    return fx * sx, fy * sy, cx * sx, cy * sy


def project_points(pc, c2w, fx, fy, cx, cy):
    """World points -> (uv [N,2], x, y, z) in the camera frame (reference utils.py:99-149)."""
    rt = c2w[:3, :3].t()
    cam = pc @ rt.t() - rt @ c2w[:3, 3]
    x, y, z = cam[:, 0], cam[:, 1], cam[:, 2]
    return torch.stack([fx * x / z + cx, fy * y / z + cy], -1), x, y, z

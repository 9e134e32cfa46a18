"""Streaming form of the reference's clustering / centre loop (smartVidCrop.py:2359-2373, :2402-2414).

The reference walks the saliency maps of a video one by one: filter map i, then -- next to a cut -- blend it into map
i+1 before that one is filtered.  `ops.Engine.cluster_center_` honours the dependency inside ONE call by rounds (a
blended map runs one round after its predecessor), which costs a call that starts a shot three or four serial tail
rounds, most of them for a single map.  `StreamPipeline` feeds a sequence of maps in calls of `batch` maps and spreads
every blend chain over CONSECUTIVE calls instead: a call processes the maps whose predecessor is final, leaves the rest
of their chains untouched (SVC_MAP_HELD, include/svc.h) and picks them up -- behind their now final predecessor -- in the
next call, so a call has ONE tail round while maps and centres stay those of the reference's loop
(tests/test_gpu_pipeline.py::test_stream_pipeline_equals_the_oracle_loop).  The maps live in one device ring and are
never copied except when the ring wraps.

Host logic only (`plan_call` is pure Python, tested without a GPU); device work goes through ops.Engine."""
import numpy as np

BLEND_NEXT, MAP_HELD = 1, 2        # include/svc.h
FINAL, RAW = 0, 1


def plan_call(states, bnext, flush=False):
    """One call over a span of maps in stream order.  states[i]: FINAL (already filtered) or RAW; bnext[i]: map i is
    blended into map i+1 once it is final.  -> (flags uint8[n] for svc_cluster_center, processed indices).
    flush: let the call run every chain to its end in rounds instead of deferring."""
    n = len(states)
    flags = np.zeros(n, np.uint8)
    done = []
    for i in range(n):
        nxt_raw = i + 1 < n and states[i + 1] == RAW
        if states[i] == FINAL:
            flags[i] = MAP_HELD | (BLEND_NEXT if (bnext[i] and nxt_raw) else 0)
            continue
        ready = i == 0 or not bnext[i - 1] or states[i - 1] == FINAL
        if flush:
            flags[i] = BLEND_NEXT if (bnext[i] and nxt_raw) else 0
            done.append(i)
        elif ready:
            done.append(i)              # its own blend into i+1 happens in the call that processes i+1
        else:
            flags[i] = MAP_HELD
    return flags, done


class StreamPipeline:
    """frames / maps in, centres out, one tail round per call.  Not thread safe; one HIP stream (the current one at
    construction, or `stream`)."""

    def __init__(self, engine, CP, sal_h, sal_w, batch=32, ring_batches=8, max_span_batches=3, stream=None, depth=2,
                 maps_out=None, timing=False, ring=None):
        """maps_out: uint8 CUDA [>= number of maps, h, w]: every filtered map is copied to maps_out[stream index] when
        it is final (VD['smaps']).  timing: HIP events around the network and tail phases of every submit_frames call
        (phase_ms()).  ring: the caller's own map storage, uint8 CUDA [rows, h, w] holding the WHOLE stream (row = stream
        index): nothing wraps, a finished map stays where it is (the multi-video scheduler, retargetvid_amd/scheduler.py:
        a video's filtered maps are a slice of it), and a call may bring more than `batch` maps."""
        import torch
        self.eng, self.CP, self.h, self.w, self.batch = engine, CP, int(sal_h), int(sal_w), int(batch)
        self.dev = engine.device
        self.stream = stream if stream is not None else torch.cuda.current_stream(self.dev)
        self.max_span = int(max_span_batches) * self.batch
        self.external = ring is not None
        if self.external:
            assert ring.is_cuda and ring.dtype == torch.uint8 and tuple(ring.shape[1:]) == (self.h, self.w) and ring.is_contiguous()
            self.ring, self.cap = ring, int(ring.shape[0])
            xy_rows = self.max_span + 2 * self.batch + 64     # a call's span: what is carried + what it brings
        else:
            self.cap = int(ring_batches) * self.batch
            assert self.cap >= self.max_span + self.batch
            self.ring = torch.empty((self.cap, self.h, self.w), dtype=torch.uint8, device=self.dev)
            xy_rows = self.cap
        self.depth = max(1, int(depth))
        self._xy = [torch.empty((xy_rows, 2), dtype=torch.float64).pin_memory() for _ in range(self.depth + 1)]
        self._ev = [torch.cuda.Event() for _ in range(self.depth + 1)]
        self.maps_out = maps_out
        self.timing = bool(timing)
        self._tev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(self.depth + 1)] if timing else None
        self._phase = [0.0, 0.0, 0]
        self.reset()

    def reset(self):
        self.base = 0               # ring row of the first map of the span
        self.gid0 = 0               # stream index of that map
        self.states, self.bnext = [], []      # per map of the span
        self.calls = []             # outstanding: (slot, [(row in span at call time, stream index)])
        self.n_calls = 0
        self.submitted = 0

    # ---- feeding ------------------------------------------------------------------------------------------------
    def slot_for(self, n):
        """Ring rows for the next n maps (the caller may write raw saliency maps there itself: `out=` of ops.saliency)."""
        import torch
        end = self.base + len(self.states)
        if self.external:
            if n <= 0 or end + n > self.cap or n > 2 * self.batch + 64:
                raise ValueError('StreamPipeline: %d maps do not fit the caller\'s map storage (row %d of %d)' % (n, end, self.cap))
            return self.ring[end:end + n]
        assert 0 < n <= self.batch
        if end + n > self.cap:                          # wrap: the open span moves to the front of the ring
            k = len(self.states)
            assert k + n <= self.cap
            with torch.cuda.stream(self.stream):
                if k:
                    tmp = self.ring[self.base:end].clone() if self.base < k else None
                    self.ring[:k].copy_(tmp if tmp is not None else self.ring[self.base:end])
            self.base, end = 0, k
        return self.ring[end:end + n]

    def submit_frames(self, frames, blend_next, small=False):
        """frames: uint8 CUDA [n,H,W,3] (full size: down-scaled here; small=True: already at saliency size).  Runs
        down-scale -> UNISAL saliency -> threshold -> one clustering round.  blend_next[n]: the cut test per map."""
        import torch
        n = int(frames.shape[0])
        dst = self.slot_for(n)
        tev = self._tev[self.n_calls % (self.depth + 1)] if self.timing else None
        with torch.cuda.stream(self.stream):
            if tev:
                tev[0].record(self.stream)
            sm = frames if small else self.eng.resize_frames(frames, self.h, self.w)
            self.eng.saliency(sm, out=dst, threshold=self.CP['t_threshold'])       # (threshold fused into the network's last kernel)
            if tev:
                tev[1].record(self.stream)
        return self._call(n, blend_next, timed=bool(tev))

    def submit_maps(self, maps, blend_next):
        """maps: uint8 CUDA [n,h,w], already thresholded."""
        import torch
        n = int(maps.shape[0])
        dst = self.slot_for(n)
        with torch.cuda.stream(self.stream):
            dst.copy_(maps)
        return self._call(n, blend_next)

    def submit_rows(self, n, blend_next):
        """The caller has written n thresholded maps into slot_for(n) on this pipeline's stream itself."""
        return self._call(int(n), blend_next)

    def _call(self, n_new, blend_next, flush=False, timed=False):
        import torch
        if len(self.calls) >= self.depth:               # checked before any state changes: collect() and retry is safe
            raise RuntimeError('StreamPipeline: collect() the oldest call before submitting another (depth %d)' % self.depth)
        if n_new:
            bn = [bool(v) for v in blend_next]
            assert len(bn) == n_new
            self.states += [RAW] * n_new
            self.bnext += bn
            self.submitted += n_new
        if not self.states:
            return None
        if len(self.states) > self.max_span:
            flush = True                                # chains too long to carry: this call runs them out in rounds
        flags, done = plan_call(self.states, self.bnext, flush)
        k = len(self.states)
        slot = self.n_calls % (self.depth + 1)
        with torch.cuda.stream(self.stream):
            xy = self.eng.cluster_center_(self.ring[self.base:self.base + k], flags, self.CP)
            self._xy[slot][:k].copy_(xy, non_blocking=True)
            if self.maps_out is not None and done:      # runs of finished maps -> their place in the caller's tensor
                r0 = 0
                while r0 < len(done):
                    r1 = r0
                    while r1 + 1 < len(done) and done[r1 + 1] == done[r1] + 1:
                        r1 += 1
                    a, b = done[r0], done[r1] + 1
                    self.maps_out[self.gid0 + a:self.gid0 + b].copy_(self.ring[self.base + a:self.base + b])
                    r0 = r1 + 1
            if timed:
                self._tev[slot][2].record(self.stream)
            self._ev[slot].record(self.stream)
        self.calls.append((slot, [(i, self.gid0 + i) for i in done], timed))
        self.n_calls += 1
        for i in done:
            self.states[i] = FINAL
        # the span the next call starts from: the first map that is still raw, or final with a successor to blend into
        keep = k
        for i in range(k):
            if self.states[i] == RAW or (self.bnext[i] and (i + 1 == k or self.states[i + 1] == RAW)):
                keep = i
                break
        self.base += keep
        self.gid0 += keep
        del self.states[:keep], self.bnext[:keep]
        return len(done)

    # ---- results ------------------------------------------------------------------------------------------------
    def collect(self):
        """Waits for the oldest outstanding call -> [(stream index, x, y)] of the maps it finished (NaN = no centre)."""
        if not self.calls:
            return []
        slot, rows, timed = self.calls.pop(0)
        self._ev[slot].synchronize()
        if timed:
            t = self._tev[slot]
            self._phase[0] += t[0].elapsed_time(t[1])
            self._phase[1] += t[1].elapsed_time(t[2])
            self._phase[2] += 1
        xy = self._xy[slot].numpy()
        return [(g, float(xy[i, 0]), float(xy[i, 1])) for i, g in rows]

    def collect_arrays(self):
        """collect() without the per-map Python objects -> (stream indices int64 [m], centres float64 [m, 2]; NaN = none)."""
        if not self.calls:
            return np.empty(0, np.int64), np.empty((0, 2), np.float64)
        slot, rows, timed = self.calls.pop(0)
        self._ev[slot].synchronize()
        if timed:
            t = self._tev[slot]
            self._phase[0] += t[0].elapsed_time(t[1])
            self._phase[1] += t[1].elapsed_time(t[2])
            self._phase[2] += 1
        if not rows:
            return np.empty(0, np.int64), np.empty((0, 2), np.float64)
        r = np.asarray(rows, np.int64)
        return r[:, 1], self._xy[slot].numpy()[r[:, 0]].copy()

    def flush(self):
        """Enqueues the call that runs out whatever is still carried (in rounds) without waiting for it; -> results
        collected on the way (to make room for the call).  Several pipelines can be flushed before any is waited for."""
        out = []
        while len(self.calls) >= self.depth:
            out += self.collect()
        if any(s == RAW for s in self.states):
            self._call(0, None, flush=True)
        return out

    def finish(self):
        """flush(), then every result not collected yet.  A trailing blend flag without a successor (the stream's last
        map) is dropped, as in the reference's loop."""
        out = self.flush()
        while self.calls:
            out += self.collect()
        self.base += len(self.states)
        self.gid0 += len(self.states)
        self.states, self.bnext = [], []
        return out

    def phase_ms(self, reset=True):
        """-> (network ms, tail ms) per timed call since the last reset (timing=True)."""
        n = max(self._phase[2], 1)
        out = (self._phase[0] / n, self._phase[1] / n)
        if reset:
            self._phase = [0.0, 0.0, 0]
        return out

    def final_rows(self, first, count):
        """Ring rows of `count` maps starting at stream index `first`, if they are still in the ring un-wrapped (the
        caller copies filtered maps out right after collect())."""
        r0 = self.base - (self.gid0 - first)
        assert r0 >= 0 and r0 + count <= self.cap
        return self.ring[r0:r0 + count]

"""Job-level scheduler: MANY videos through the saliency-to-crop path as ONE stream of frames per HIP stream.

The reference's driver loop (smartVidCrop.py:2722-2790) crops one video after the other, and so did this package's
crop_videos: a video of 105 selected frames is 32 + 32 + 32 + 9 frames for the network (the last piece 28 % full), every
video pays its own shot-start tail rounds, and its host stages run between two videos' device work.  Here the selected
frames of consecutive videos form one stream per lane (lane = engine + HIP stream):

  * the network always gets full chunks of `chunk` frames, packed ACROSS video boundaries (a frame's map does not depend
    on its neighbours in the chunk);
  * the maps of a lane live in one device tensor in stream order; a video boundary is just a shot start: the cut-blend
    flags of a video end with two zeros (smartVidCrop.py:2324-2327: `i < fc_sel - 2`), so nothing is blended across it,
    and the all-zero map the reference's off-by-one gives the last selected frame of every read batch (:408-453) is a row
    the network never writes;
  * threshold, cluster filter, cut blend and centres follow every chunk on the lane's stream through
    pipeline.StreamPipeline (ONE tail round per call, blend chains carried over to the next call), whatever the videos
    in the chunk;
  * when the last centre of a video has arrived its host stages (smartVidCrop.after_ingest: empty-centre fill,
    interpolation, low-pass, LOESS, boxes -- native code behind temporal.py, which releases the interpreter lock) run on
    a small thread pool while the feeder keeps the lanes busy.

One feeder thread drives all lanes round-robin (a lane takes the next video from the common queue when it runs low), so
the low-occupancy tail of one lane overlaps the full-chip network of the others, as bench.py's batches do.  Every video's
result equals smartVidCrop.smart_vid_crop_ratios on that video alone (tests/test_gpu_scheduler.py).

Videos given as CALLABLES are built on demand.  Without shot_net the feeder thread calls them, one at a time, when a lane takes
the video.  With shot_net= the PLANNER threads call them (shot detection runs ahead of the lanes): up to JobScheduler.PLANNERS
callables run concurrently, on non-feeder threads and with a planner's HIP stream current -- they must be thread-safe -- and
at most `plan_ahead` videos (default: lanes + 2 per planner) are materialised beyond the one the feeder took last, so a job of
thousands of on-demand videos holds a bounded number of them in memory.

Host logic + torch plumbing only; device work goes through ops.Engine (the C ABI)."""
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

from . import smartVidCrop as S


_LANE_STREAMS = {}          # device index -> HIP streams of the lanes, created once per process


def lane_streams(dev, n):
    """The lanes' HIP streams, kept for the life of the process: a scheduler made per job must not walk through the
    runtime's hardware queues (GPU_MAX_HW_QUEUES) -- measured: the job whose streams were the 13th..16th of the process ran
    10 x slower than its neighbours (tools/time_scheduler.py --fresh 1)."""
    import torch
    pool = _LANE_STREAMS.setdefault(dev.index, [])
    while len(pool) < n:
        pool.append(torch.cuda.Stream(device=dev))
    return pool[:n]


def lane_rows_for(videos, CP, lanes, cap=4096):
    """Rows (selected frames) of a lane's frame and map storage for this job: what its share of the job needs (twice, so that
    an uneven split does not force a replacement), never less than a chunk's worth, never more than ``cap``; a job of
    callables (videos built on demand: sizes unknown) gets ``cap``.  A video longer than this still gets its own storage
    (_Lane._alloc takes the larger of the two)."""
    n = []
    for v in videos:
        if callable(v):
            return cap
        fc = int(v.get('frame_count', 0) or len(v['frames']))
        n.append(fc // max(1, int(CP['skip'])) + 3)
    if not n:
        return 128
    return int(min(cap, max(128, 2 * (sum(n) // max(1, lanes) + 1))))


class LookAhead:
    """Back-pressure between the planner threads (which build on-demand videos and run shot detection AHEAD of the lanes) and the
    feeder (which takes the videos in order): item i may be produced only while i < taken + limit, `taken` = how many items the
    consumer has asked for so far.  The consumer only ever waits for item taken - 1, which the condition always admits, so the
    two sides cannot dead-lock; stop() and fail() release every waiter.  Pure host logic (tests/test_host_logic.py)."""

    def __init__(self, limit):
        import threading
        self.limit = max(1, int(limit))
        self.taken = 0
        self.stopped = False
        self.error = None
        self.high_water = 0                # most items ever admitted beyond the consumer's position (<= limit)
        self._cv = threading.Condition()

    def admit(self, i, poll=0.5):
        """Producer side: blocks until item i may be produced.  -> False when the job was stopped or has failed."""
        with self._cv:
            while not (self.stopped or self.error is not None or i < self.taken + self.limit):
                self._cv.wait(poll)
            if self.stopped or self.error is not None:
                return False
            self.high_water = max(self.high_water, i - self.taken + 1)
            return True

    def take(self):
        """Consumer side: the next item is being asked for."""
        with self._cv:
            self.taken += 1
            self._cv.notify_all()

    def stop(self):
        with self._cv:
            self.stopped = True
            self._cv.notify_all()

    def fail(self, err):
        with self._cv:
            if self.error is None:
                self.error = err
            self._cv.notify_all()


class _Video:
    __slots__ = ('idx', 'video', 'plan', 'lane', 'row0', 'xy', 'remaining', 'maps', 'pos', 'done', 'sink', 'ready', 'census')

    def __init__(self, idx, video, plan, sink):
        self.idx, self.video, self.plan = idx, video, plan
        self.sink = sink                   # the result list of the job this video belongs to (a straggler of a failed job never writes into the next job's)
        self.xy = np.full((plan['n_sel'], 2), np.nan)
        self.remaining = plan['n_sel']
        self.pos = 0                       # next selected frame to take in
        self.maps = None
        self.done = False


class _Lane:
    """One engine + HIP stream + map storage; frames and maps of consecutive videos in stream order."""

    def __init__(self, sched, engine, stream, k):
        self.sched, self.eng, self.stream, self.k = sched, engine, stream, k
        self.geom = None
        self.cur = None                    # video being taken in
        self.videos = {}                   # slot -> _Video with maps in this lane's storage
        self.next_slot = 0
        self.pipe = None
        self.exhausted = False
        self.cap = 0                       # no storage yet: the first video allocates it
        self.frames_in = self.frames_done = self.rows_in = self.rows_called = 0

    # ---- storage ----------------------------------------------------------------------------------------------------
    def _alloc(self, sal_h, sal_w, min_rows):
        import torch
        from . import pipeline as _pl
        sc = self.sched
        rows = max(sc.lane_rows, min_rows)
        self.geom = (sal_h, sal_w)
        dev = self.eng.device
        with torch.cuda.stream(self.stream):
            self.small = torch.empty((rows, sal_h, sal_w, 3), dtype=torch.uint8, device=dev)     # network input frames, in order
            self.maps = torch.zeros((rows, sal_h, sal_w), dtype=torch.uint8, device=dev)         # stream rows (zero rows stay zero)
            self.tmp = torch.empty((sc.chunk, sal_h, sal_w), dtype=torch.uint8, device=dev)
            # threshold census (the regime diagnostic, per video): pixels of a row's raw map at t - 1, t, t + 1, added by the network's
            # last kernel (svc_saliency_census_u8); rows without a network pass stay 0
            self.census = torch.zeros((rows, 4), dtype=torch.int32, device=dev)
            self.census_tmp = torch.zeros((sc.chunk, 4), dtype=torch.int32, device=dev)
        self.cap = rows
        self.row_of_frame = np.empty(rows, np.int64)
        self.flags = np.zeros(rows, np.uint8)
        self.vid_of_row = np.empty(rows, np.int32)
        self.local_of_row = np.empty(rows, np.int32)
        self.frames_in = self.frames_done = self.rows_in = self.rows_called = 0
        self.pipe = _pl.StreamPipeline(self.eng, sc.CP, sal_h, sal_w, batch=sc.chunk, stream=self.stream, depth=sc.depth,
                                       ring=self.maps)

    # ---- intake -----------------------------------------------------------------------------------------------------
    def _start_video(self, v):
        """Room for the whole video in this lane's storage (a video never straddles two storages: its blend chains and
        its filtered maps stay in one)."""
        plan = v.plan
        geom = (plan['sal_h'], plan['sal_w'])
        if self.geom != geom or self.rows_in + plan['n_sel'] > self.cap:
            self.drain()
            self._alloc(geom[0], geom[1], plan['n_sel'])
        v.lane, v.row0 = self, self.rows_in
        v.maps = self.maps[v.row0:v.row0 + plan['n_sel']]         # a VIEW of the lane's storage until the video's last centre has arrived (_dispatch clones it then)
        slot = self.next_slot
        self.next_slot += 1
        self.videos[slot] = v
        n = plan['n_sel']
        r0 = self.rows_in
        self.flags[r0:r0 + n] = plan['flags'] if plan['flags'] is not None else 0
        self.vid_of_row[r0:r0 + n] = slot
        self.local_of_row[r0:r0 + n] = np.arange(n, dtype=np.int32)
        self.cur = v

    def _take_piece(self):
        """The next piece of the current video: down-scaled frames behind the lane's frames, their rows behind its rows."""
        import torch
        v, plan, sc = self.cur, self.cur.plan, self.sched
        frames = v.video['frames']
        n = plan['n_sel']
        per = max(sc.chunk, min(sc.piece_frames, sc.piece_bytes // max(1, plan['h'] * plan['w'] * 3)))
        m = min(per, n - v.pos)
        loc = np.arange(v.pos, v.pos + m)
        net = loc[~plan['zero_map'][v.pos:v.pos + m]]
        if len(net):
            idx = [plan['true_inds'][j] for j in net]
            with torch.cuda.stream(self.stream):
                small = S._small_frames(self.eng, frames, idx, plan['sal_h'], plan['sal_w'], self.eng.device)
                self.small[self.frames_in:self.frames_in + len(net)].copy_(small)
            self.row_of_frame[self.frames_in:self.frames_in + len(net)] = v.row0 + net
            self.frames_in += len(net)
        self.rows_in += m
        v.pos += m
        if v.pos >= n:
            self.cur = None

    def _fill(self):
        """Frames for at least one full chunk, as long as the job has any."""
        sc = self.sched
        while self.frames_in - self.frames_done < sc.chunk:
            if self.cur is None:
                t = time.perf_counter()
                v = sc._next_video(self)
                sc.host_s['plan'] += time.perf_counter() - t
                if v is None:
                    self.exhausted = True
                    return
                self._start_video(v)
            t = time.perf_counter()
            self._take_piece()
            sc.host_s['intake'] += time.perf_counter() - t

    # ---- one chunk --------------------------------------------------------------------------------------------------
    def step(self):
        """Takes in what is needed and enqueues one network chunk + its tail call.  -> False when the lane has nothing left."""
        import torch
        sc = self.sched
        if not self.exhausted:
            self._fill()
        if self.pipe is None:
            return False
        k = min(sc.chunk, self.frames_in - self.frames_done)
        if k == 0 and self.rows_in == self.rows_called:
            return False
        if len(self.pipe.calls) >= self.pipe.depth:
            t = time.perf_counter()
            res = self.pipe.collect_arrays()
            sc.host_s['wait'] += time.perf_counter() - t
            sc._dispatch(self, *res)
        t = time.perf_counter()
        f0 = self.frames_done
        # rows of this call: up to (not including) the row of the next frame the network has not seen
        R = int(self.row_of_frame[f0 + k]) if f0 + k < self.frames_in else self.rows_in
        # Rows that carry no frame (the all-zero last map of every read batch, videos of one selected frame, read_batch = 1)
        # can pile up between two frames: a call takes at most what the tail's storage holds, the rest goes to the next
        # call(s) -- a degenerate video costs calls, it does not abort the job
        limit = 2 * sc.chunk + 64
        if R - self.rows_called > limit:
            R = self.rows_called + limit
            k = int(np.searchsorted(self.row_of_frame[f0:f0 + k], R))    # frames whose rows are inside the call
        n_rows = R - self.rows_called
        self.pipe.slot_for(n_rows)                                   # (checks that the rows fit the storage)
        with torch.cuda.stream(self.stream):
            if k:
                rows = self.row_of_frame[f0:f0 + k]
                r0 = int(rows[0])
                cen = 2 <= int(sc.CP['t_threshold']) <= 254              # (t = 1: level t - 1 is the value of the rows without a network pass)
                if int(rows[-1]) - r0 + 1 == k:                       # no zero row inside: the network writes in place
                    self.eng.saliency(self.small[f0:f0 + k], out=self.maps[r0:r0 + k], threshold=sc.CP['t_threshold'],
                                      census=self.census[r0:r0 + k] if cen else None)
                else:
                    if cen:
                        self.census_tmp[:k].zero_()
                    self.eng.saliency(self.small[f0:f0 + k], out=self.tmp[:k], threshold=sc.CP['t_threshold'],
                                      census=self.census_tmp[:k] if cen else None)
                    brk = np.flatnonzero(np.diff(rows) != 1) + 1
                    a = 0
                    for b in list(brk) + [k]:                         # runs of consecutive rows
                        ra = int(rows[a])
                        self.maps[ra:ra + (b - a)].copy_(self.tmp[a:b])
                        if cen:
                            self.census[ra:ra + (b - a)].copy_(self.census_tmp[a:b])
                        a = int(b)
        t1 = time.perf_counter()
        self.pipe.submit_rows(n_rows, self.flags[self.rows_called:R])
        t2 = time.perf_counter()
        sc.host_s['enqueue_net'] += t1 - t
        sc.host_s['enqueue_tail'] += t2 - t1
        self.frames_done += k
        self.rows_called = R
        sc.n_chunks += 1
        sc.n_net_frames += k
        return True

    def flush(self):
        """Enqueues the call that runs out what is still carried, without waiting for it."""
        p = self.pipe
        if p is not None:
            while len(p.calls) >= p.depth:
                self.sched._dispatch(self, *p.collect_arrays())
            p.flush()

    def finish(self):
        if self.pipe is not None:
            self.flush()
            while self.pipe.calls:
                self.sched._dispatch(self, *self.pipe.collect_arrays())
            self.pipe.finish()

    def drain(self):
        """Runs out everything this lane holds (its storage is about to be replaced)."""
        if self.pipe is None:
            return
        while self.frames_in > self.frames_done or self.rows_in > self.rows_called:
            ex, self.exhausted = self.exhausted, True         # no intake while draining
            self.step()
            self.exhausted = ex
        self.finish()
        self.pipe = None
        self.geom = None


class JobScheduler:
    """crop_videos' engine room.  ``videos``: sequence of ingest_pickle dicts or zero-argument callables producing them."""

    def __init__(self, CP, ratios=None, lanes=4, chunk=32, state_dict=None, seed=0, engines=None, shot_net=None,
                 host_threads=3, depth=2, lane_rows=4096, piece_frames=256, piece_bytes=256 << 20, plan_ahead=None):
        import torch
        from . import ops as _ops
        if CP['exit_on_spread_sal'] or CP['exit_on_low_cvrg'] or CP['t_border'] != -1:
            raise NotImplementedError('mean-saliency / coverage gates and border detection are disabled in both '
                                      'published parameter sets and are not part of this path')
        self.CP = CP
        self.ratios = tuple(ratios) if ratios else (CP['out_ratio'],)
        self.chunk, self.depth, self.lane_rows = int(chunk), int(depth), int(lane_rows)
        self.piece_frames, self.piece_bytes = int(piece_frames), int(piece_bytes)
        self.shot_net = shot_net
        self.plan_ahead = None if plan_ahead is None else max(1, int(plan_ahead))    # None: lanes + 2 per planner (set in _start_planner)
        self._plan_ready = self._planned = self._plan_err = None
        self._plan_nets = []                                  # the planner threads' networks: shot_net and its clones (made by the first job)
        self.dev = torch.device('cuda', torch.cuda.current_device())
        self.own_engines = engines is None
        if engines is None:
            if state_dict is None:                               # built once, not once per lane
                from . import weights as _weights
                state_dict = _weights.make_synthetic_state_dict(seed)
            engines = [_ops.Engine(state_dict, device=self.dev.index, seed=seed) for _ in range(max(1, int(lanes)))]
        self.engines = list(engines)
        self.streams = lane_streams(self.dev, len(self.engines))
        self.pool = ThreadPoolExecutor(max_workers=max(1, int(host_threads)))

    def close(self):
        self.pool.shutdown(wait=True)
        for n in self._plan_nets:                              # the clones are the scheduler's; shot_net itself is the caller's
            if n is not self.shot_net:
                n.close()
        self._plan_nets = []
        if self.own_engines:
            for e in self.engines:
                e.close()
        self.engines = []

    # ---- the job ------------------------------------------------------------------------------------------------------
    def run(self, videos):
        """-> list, in input order, of {ratio: (VD, smart_crop_results)}; each entry equals smart_vid_crop_ratios on that
        video alone.  stats of the run in self.stats."""
        import torch
        if not self.CP['clust_filt']:
            raise NotImplementedError('JobScheduler runs the cluster filter inside the stream (clust_filt=True, as in both '
                                      'published parameter sets); use crop_videos(..., packed=False) otherwise')
        self.videos = list(videos)
        self.out = [None] * len(self.videos)
        self.next_idx = 0
        self.futures = []
        self.n_chunks = self.n_net_frames = 0
        self.host_s = dict(plan=0.0, intake=0.0, enqueue_net=0.0, enqueue_tail=0.0, wait=0.0, dispatch=0.0)
        t0 = time.perf_counter()
        lanes = [_Lane(self, e, s, k) for k, (e, s) in enumerate(zip(self.engines, self.streams))]
        t0 = time.perf_counter()
        planner = self._start_planner()
        try:
            with torch.cuda.device(self.dev):
                live = list(lanes)
                while live:
                    live = [ln for ln in live if ln.step()]
                for ln in lanes:                              # every lane's last call is enqueued before any is waited for
                    ln.flush()
                for ln in lanes:
                    ln.finish()
        except BaseException:
            self._stop_planner(planner)
            # the feeder failed (a bad video in plan_video, rows that do not fit): no host-stage task of this job may still be
            # running when the caller sees the error -- the scheduler is reused for the next job
            for f in self.futures:
                f.cancel()
            for f in self.futures:
                try:
                    f.result()
                except BaseException:
                    pass
            torch.cuda.synchronize(self.dev)
            raise
        self._stop_planner(planner)
        t1 = time.perf_counter()
        err = None
        for f in self.futures:
            try:
                f.result()
            except BaseException as e:                        # surfaced in the caller's thread
                err = err or e
        if err is not None:
            raise err
        t2 = time.perf_counter()
        missing = [i for i, o in enumerate(self.out) if o is None]
        if missing:
            raise RuntimeError('JobScheduler: videos %r were never completed' % (missing[:8],))
        self.stats = dict(videos=len(self.videos), chunks=self.n_chunks, network_frames=self.n_net_frames,
                          mean_chunk_fill=self.n_net_frames / max(1, self.n_chunks) / self.chunk,
                          seconds_device_side=t1 - t0, seconds_host_stage_drain=t2 - t1, lanes=len(lanes),
                          feeder_seconds={k: round(v, 4) for k, v in self.host_s.items()})
        if planner:                                           # shot detection ahead of the lanes: how far ahead it was allowed / ever got
            self.stats.update(planners=len(planner), plan_ahead=self.plan_ahead, plan_high_water=self._look.high_water)
        return self.out

    # ---- feeder-side helpers ------------------------------------------------------------------------------------------
    def _next_video(self, lane):
        import torch
        if self.next_idx >= len(self.videos):
            return None
        i = self.next_idx
        shots = None
        if self._plan_ready is not None:                      # shot detection ran (or is running) in the planner thread
            self.next_idx += 1
            self._look.take()                                 # the planners may now materialise video i + plan_ahead
            self._plan_ready[i].wait()
            if self._plan_err is not None:
                raise self._plan_err
            v, shots = self._planned[i]
            self._planned[i] = None
        else:
            self.next_idx += 1
        with torch.cuda.stream(lane.stream):
            if self._plan_ready is None:
                v = self.videos[i]() if callable(self.videos[i]) else self.videos[i]
            plan = S.plan_video(v, self.CP, engine=lane.eng, shot_net=self.shot_net, shots=shots)
        return _Video(i, v, plan, self.out)

    # Shot detection ahead of the lanes (jobs with shot_net=): TransNet is the longest device stage of the video path, and run
    # from the feeder (inside plan_video) every video's windows, their copy to the host and the scene walk sat between two
    # lanes' enqueues.  One thread walks the videos in order on a stream of its own with the network's own engine; the
    # feeder only picks up the finished (video, shots) pairs, so the saliency lanes overlap with the shot network.
    # Three planner threads (the videos go round them), the second and third with a clone of the network on an engine of its own:
    # one thread's copy of the probabilities to the host and its scene walk overlap with the others' windows on the device
    # (one thread: 10.3 ms per video of which 6.8 ms is the network; the 200-video job 2.46 s with one, 2.25 - 2.27 with two,
    # 2.17 - 2.19 with three, 2.28 with four: three alternations).
    PLANNERS = 3

    def _start_planner(self):
        import threading
        self._plan_ready, self._planned, self._plan_err, self._plan_stop = None, None, None, False
        if self.shot_net is None or not self.videos:
            return None
        self._plan_ready = [threading.Event() for _ in self.videos]
        self._planned = [None] * len(self.videos)
        while len(self._plan_nets) < min(self.PLANNERS, len(self.videos)):
            # planner 0 runs the caller's network only when that network owns its engine: a network built on an engine the
            # caller also uses elsewhere (a lane's, get_engine()'s) must not be driven from a second thread -- a handle's
            # workspace serves one call at a time -- so it is cloned for planner 0 as well
            own = bool(getattr(self.shot_net, '_own', False))
            self._plan_nets.append(self.shot_net if (not self._plan_nets and own) else self.shot_net.clone())
        if self.plan_ahead is None:
            self.plan_ahead = len(self.engines) + 2 * len(self._plan_nets)
        self._look = LookAhead(self.plan_ahead)
        ths = [threading.Thread(target=self._plan_ahead, args=(k, len(self._plan_nets)), name='svc-shot-planner-%d' % k, daemon=True)
               for k in range(len(self._plan_nets))]
        for th in ths:
            th.start()
        return ths

    def _plan_ahead(self, k, stride):
        import torch
        net = self._plan_nets[k]
        try:
            with torch.cuda.device(self.dev):
                st = torch.cuda.Stream(device=self.dev)
                with torch.cuda.stream(st):
                    for i in range(k, len(self.videos), stride):
                        # back-pressure: video i is materialised only when the feeder is within plan_ahead videos of it (the
                        # planners take ~3 ms per video, the lanes ~11: unbounded, a job of on-demand callables would be built
                        # whole long before the lanes reach it).  The feeder only ever waits for video next_idx - 1, which this
                        # condition always admits: no deadlock.
                        if not self._look.admit(i) or self._plan_stop or self._plan_err is not None:
                            break
                        v = self.videos[i]() if callable(self.videos[i]) else self.videos[i]
                        shots = None
                        if v.get('trans_inds') is None:
                            shots = S.detect_shots(v['frames'], v['fr'], self.CP, net=net, engine=net.eng, trans_threshold=S.TRANS_THRESHOLD)
                        else:
                            st.synchronize()                  # whatever the callable enqueued is done before a lane reads the frames
                        self._planned[i] = (v, shots)
                        self._plan_ready[i].set()
        except BaseException as e:                            # surfaced by the feeder (_next_video)
            self._plan_err = e
            self._look.fail(e)
        finally:
            for ev in self._plan_ready[k::stride]:
                ev.set()
            if self._plan_err is not None:                    # nobody waits for a video the other thread will not reach
                for ev in self._plan_ready:
                    ev.set()

    def _stop_planner(self, ths):
        if ths:
            self._plan_stop = True
            self._look.stop()
            for th in ths:
                th.join()

    def _dispatch(self, lane, gids, xy):
        """Centres of finished maps -> their videos; a video whose last centre arrived goes to the host-stage pool."""
        import torch
        if len(gids) == 0:
            return
        t = time.perf_counter()
        slots = lane.vid_of_row[gids]
        loc = lane.local_of_row[gids]
        for slot in np.unique(slots):
            m = slots == slot
            v = lane.videos[int(slot)]
            v.xy[loc[m]] = xy[m]
            v.remaining -= int(m.sum())
            if v.remaining == 0 and not v.done:
                v.done = True
                del lane.videos[int(slot)]
                # the video's filtered maps leave the lane's storage: a result that is kept does not keep rows x h x w bytes
                # (and whatever storage drain() has replaced since) alive, and results of different videos do not alias
                with torch.cuda.stream(lane.stream):
                    v.maps = v.maps.clone()
                    n_sel = v.plan['n_sel']
                    v.census = lane.census[v.row0:v.row0 + n_sel].sum(0)      # the video's own rows (read on the host by _finish_video)
                    v.ready = torch.cuda.Event()
                    v.ready.record(lane.stream)               # _finish_video waits for the copy before anything reads it
                self.futures.append(self.pool.submit(self._finish_video, v))
        self.host_s['dispatch'] += time.perf_counter() - t

    def _finish_video(self, v):
        """Host stages of one video (pool thread)."""
        S.sc_init_time()
        v.ready.synchronize()              # the video's own copy of its maps is complete (made on the lane's stream; read on any)
        VD = S._LazySmaps(S._ingest_dict(v.plan, v.maps, xy_stream=v.xy))
        # the regime diagnostic of THIS video (smartVidCrop.after_ingest reports it): its maps' pixels at t - 1, t, t + 1 per map and level
        n_net = int((~v.plan['zero_map']).sum())
        tt = int(self.CP['t_threshold'])
        if n_net and 2 <= tt <= 254:
            c = v.census.cpu().numpy()
            VD['pixels_per_grey_level_at_threshold'] = float(c[0] + c[1] + c[2]) / (3.0 * n_net)
        out = {}
        base = None
        for ratio in self.ratios:
            cp = dict(self.CP, out_ratio=ratio)
            if base is None:
                base = S.after_ingest(VD, cp, v.lane.eng)
                out[ratio] = base
            else:
                out[ratio] = S.other_ratio(base, cp)
        v.sink[v.idx] = out
        v.video = None                     # the frames are not needed any more

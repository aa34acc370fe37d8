"""
The producer / consumer overlap of `mp train` (mpunet/train/trainer.py:238-257: `model.fit(train, workers=5,
max_queue_size=5)` -- five host threads cut batches while the GPU trains).

Here the producer is the GPU plane sampler itself (data.TrainSampler), so the overlap is between two HIP streams of one
process instead of between threads: batch i+1 is cut on a side stream (chosen by measurement: pick_side_streams ranks candidates
with a fill probe, the first four then serve ten training steps each and the one with the fastest loop is kept) while train
step i runs on the main stream. The sampler's per-candidate host read (8 bytes of accept / reject statistics) synchronises only the side stream, so the
host never waits for the train step; the step itself is one HIP-graph replay (UNet.make_graphed_train_step on fixed input
tensors -- the cut batch is copied into them, 1 MB) and the loss is accumulated on the device and read once per epoch.

  main stream : [copy b_i -> graph inputs][step i .............][copy b_i+1][step i+1 ..........]
  side stream :          [cut b_i+1: ~16 candidates, each sample + stats + 8-byte read]   [cut b_i+2 ...]
  host        : replay(i) | sampler loop of b_i+1 (blocks on the side stream only) | replay(i+1) | ...

With data parallelism (model._grad_hook set) the step stays eager (the RCCL all-reduce is not captured); the sampler overlap
and the device-side loss are the same.
"""
import torch


def pick_side_stream(dev, candidates=6, busy_ms=3.0, batches=4):
    """The best candidate of pick_side_streams: (stream, its latency in microseconds)."""
    return pick_side_streams(dev, candidates, busy_ms, batches)[0]


def pick_side_streams(dev, candidates=6, busy_ms=3.0, batches=4):
    """Producer streams that really run BESIDE the current stream, best first. The HIP runtime multiplexes its streams onto a few hardware
    queues; a side stream that lands on the queue of the training stream executes behind the whole graph replay, and every host read
    of the sampler then costs a train step (measured, gpurun R5p / R5r: `train_e2e` 0.64 of the bench line when the streams created by
    the legs before it had shifted the assignment, 0.97 otherwise). So the pipeline MEASURES: the current stream is kept busy for a
    few milliseconds with large fills, each candidate stream (high priority: see below) gets one tiny kernel, and the candidate whose kernel
    completes soonest -- it did not wait for the fills -- is taken. If even the best candidate of a batch waited for a sizeable part
    of the fills (every one of them shares the busy queue), a further batch of streams is created (the runtime deals new streams
    round-robin over its queues), up to `batches` times. Returns [(stream, its latency in microseconds), ...] of the last batch,
    sorted by latency: the fill probe cannot see every way a stream can end up behind the train step (round 6, gpurun R6y: two
    pipelines of one process, both streams at 340 us in this probe, one loop at 2.53 and one at 2.99 ms per step -- the second
    producer waited a whole step at every host read), so TrainPipeline tries the first few under the real loop."""
    import time
    big = torch.empty(64 << 20, dtype=torch.float32, device=dev)           # 256 MB: ~0.1 ms per fill
    tiny = torch.zeros(64, dtype=torch.float32, device=dev)
    # one-off costs out of the way first (a fresh process: the first fill, the first kernel of a high-priority stream -- bench.py's
    # e2e leg reported a 6-ms "latency" for the stream its loop then ran fastest on)
    big.fill_(0.0)
    with torch.cuda.stream(torch.cuda.Stream(device=dev, priority=-1)):
        tiny.add_(1.0)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    big.fill_(1.0)
    torch.cuda.synchronize(dev)
    per_fill = max(time.perf_counter() - t0, 2e-5)
    nfill = int(min(200, max(8, busy_ms * 1e-3 / per_fill)))
    best, kept, ranked = None, [], []
    for _ in range(max(1, int(batches))):
        ranked = []
        # HIGH priority only (round 6, tools/round6/m_streams.py: twelve candidate streams under the real loop, three processes --
        # every high-priority stream ran the loop at the step's rate, 2.68-2.71 ms, the normal-priority ones at 2.81-3.77 ms, and the
        # fill probe below does not tell those apart: high-priority streams have hardware queues of their own, a normal-priority
        # one can share its queue with the library's optimizer stream or a branch of the replayed graph and then waits behind a
        # 0.4-ms kernel at every host read of the sampler)
        cands = [torch.cuda.Stream(device=dev, priority=-1) for k in range(candidates)]
        kept.append(cands)                                               # (alive until the choice is made: a freed stream's queue slot is reused)
        for st in cands:
            torch.cuda.synchronize(dev)
            for _ in range(nfill):
                big.fill_(1.0)                                           # the "train step" of the probe
            ev = torch.cuda.Event()
            t0 = time.perf_counter()
            with torch.cuda.stream(st):
                tiny.add_(1.0)
                ev.record(st)
            ev.synchronize()
            lat = (time.perf_counter() - t0) * 1e6
            ranked.append((st, lat))
            if best is None or lat < best[1]:
                best = (st, lat)
        if best[1] < 0.25 * nfill * per_fill * 1e6:                      # it ran beside the fills, not behind them
            break
    torch.cuda.synchronize(dev)
    del big
    ranked.sort(key=lambda c: c[1])
    if ranked[0][0] is not best[0]:                                       # (an earlier batch held the overall best)
        ranked.insert(0, best)
    return ranked


class TrainPipeline:
    CAL_STREAMS = 4        # producer-stream candidates tried under the real loop
    CAL_WINDOW = 10        # training steps each of them serves (the first two, which still consume the previous stream's batch, untimed)

    def __init__(self, model, sampler, graphed=None, overlap=True):
        self.model, self.sampler = model, sampler
        dev = model.device
        if dev.type != "cuda":
            raise RuntimeError("TrainPipeline drives the HIP path: it needs a GPU device")
        B, d = sampler.batch_size, sampler.dim
        C = int(sampler.volumes[0].n_channels)
        self.gx = torch.zeros((B, d, d, C), dtype=torch.float32, device=dev)      # the graph's input tensors
        self.gy = torch.zeros((B, d * d, 1), dtype=torch.uint8, device=dev)
        self.gw = torch.ones(B, dtype=torch.float32, device=dev)
        self.loss_sum = torch.zeros(1, dtype=torch.float64, device=dev)           # sum over steps of the step's mean loss
        self.steps_in_sum = 0
        self.graphed = (model._grad_hook is None) if graphed is None else bool(graphed)
        if self.graphed and model._grad_hook is not None:
            raise NotImplementedError("the graphed step is single-GPU (the gradient all-reduce stays eager)")
        self.overlap = bool(overlap)
        # the producer stream: the candidates the fill probe ranks best, then the first CAL_STREAMS of them under the REAL loop --
        # each serves CAL_WINDOW consecutive training steps (ordinary steps of the run: nothing is repeated or skipped), the window's
        # time per step is measured, and the fastest is kept (side_loop_ms: what each measured)
        self._cands = pick_side_streams(dev)[:self.CAL_STREAMS] if self.overlap else []
        self.side, self.side_latency_us = self._cands[0] if self._cands else (None, None)
        self._cal = {"i": 0, "n": 0, "t0": 0.0} if len(self._cands) > 1 else None
        self.side_loop_ms = []
        self._replay, self._lr = None, None
        self._pending = None

    # ---- producer ---------------------------------------------------------------------------------------------------------
    def _produce(self):
        """Cut one batch (on the side stream when overlapping). Returns (x, y, w, ready event | None)."""
        if not self.overlap:
            return self.sampler() + (None,)
        main = torch.cuda.current_stream()
        with torch.cuda.stream(self.side):
            x, y, w = self.sampler()
            ev = torch.cuda.Event()
            ev.record(self.side)
        for t in (x, y, w):                       # allocated on the side stream, read by the copy on the main stream
            t.record_stream(main)
        return x, y, w, ev

    # ---- consumer ---------------------------------------------------------------------------------------------------------
    def _launch_step(self):
        m = self.model
        if not self.graphed:
            m.train_step(self.gx, self.gy, self.gw, want_loss=bool(m.l2_reg))      # (want_loss also switches the l2 term's value on)
            self.loss_sum += m.loss_mean().double()         # (the backward pass's own mean: the value the graphed step adds)
            if m.l2_reg:
                self.loss_sum += m.reg_loss.double()
            return
        lr = float(m.optimizer_kwargs["lr"])
        if self._replay is None or lr != self._lr:      # first step, or ReduceLROnPlateau moved the rate: (re)capture
            self._replay = m.make_graphed_train_step(self.gx, self.gy, self.gw, loss_sum=self.loss_sum,
                                                     warmup=self._replay is None)
            self._lr = lr
            if self._replay.warmup_ran:                  # the capture's warm-up WAS this step (a real one, loss included)
                return
        self._replay()

    def step(self):
        """One training step on the next batch; returns nothing (the loss stays on the device: epoch_loss())."""
        if self._pending is None:
            self._pending = self._produce()
        x, y, w, ev = self._pending
        if ev is not None:
            torch.cuda.current_stream().wait_event(ev)
        self.gx.copy_(x, non_blocking=True)
        self.gy.copy_(y.reshape(self.gy.shape), non_blocking=True)
        self.gw.copy_(w, non_blocking=True)
        self._launch_step()
        self.steps_in_sum += 1
        self._pending = self._produce()                 # overlaps with the step just enqueued
        if self._cal is not None:
            self._calibrate_tick()

    def _calibrate_tick(self):
        """Called after every step while the producer stream is still being chosen (see __init__)."""
        import time
        c = self._cal
        c["n"] += 1
        if c["n"] == 2:
            torch.cuda.synchronize(self.model.device)
            c["t0"] = time.perf_counter()
        elif c["n"] >= self.CAL_WINDOW:
            torch.cuda.synchronize(self.model.device)
            self.side_loop_ms.append(round((time.perf_counter() - c["t0"]) / (self.CAL_WINDOW - 2) * 1e3, 4))
            c["i"] += 1
            c["n"] = 0
            if c["i"] < len(self._cands):
                self.side, self.side_latency_us = self._cands[c["i"]]
            else:
                k = min(range(len(self.side_loop_ms)), key=lambda j: self.side_loop_ms[j])
                self.side, self.side_latency_us = self._cands[k]
                self._cal = None                        # (the other candidates stay alive: a freed stream's queue slot is reused)

    def epoch_loss(self):
        """Mean over the steps since the last call of the step's mean weighted per-pixel loss: ONE device read."""
        n = max(1, self.steps_in_sum)
        tot = float(self.loss_sum.item())
        self.loss_sum.zero_()
        self.steps_in_sum = 0
        return tot / n

    def run_epoch(self, steps):
        for _ in range(int(steps)):
            self.step()
        return self.epoch_loss()

"""NOT PART OF THE PRODUCT (round 6 experiment, measured slower at every setting: docs/experiments.md, "Negative: starting the phase
early on a third stream"). The section of neuralsampleid_amd/functional.py that could FORK the deferred weight-gradient phase onto an
auxiliary stream as soon as both views had passed a stage boundary of backward (DEFER_FORK_AT / DEFER_FORK_WGS, DeferredWgrads.boundary,
the capped launch through max_workgroups of nsid_linear_bwd_weight_grouped). Kept verbatim for the record; to time it again put the class
back and call DEFERRED.boundary() at the end of functional.downsample_backward."""

# ------------------------------------------------------------------------------------------------ deferred weight gradients
# Nothing waits for a weight gradient until the optimiser (or the gradient all-reduce), yet launched in stream order it sits on its
# view's dependent chain: the next backward-data GEMM starts behind it. With DEFER_WGRAD the conv layers of the Grapher / FFN blocks and
# the stem only RECORD their weight-gradient problem (dr, the saved GEMM input and its pending affine stay alive); when autograd has
# run the whole backward pass (engine callback) the two view streams are joined ONCE and the recorded problems are issued together
# (ops.linear_bwd_weight_batch: both views of a layer as one problem over 2M rows, layers grouped into a few launches).
# References: the backward of every conv at encoder/gcn_lib/torch_vertex.py:152-162, encoder/graph_encoder.py:74-77,
# encoder/gcn_lib/torch_nn.py:56; train.py:70-75.
DEFER_WGRAD = 1      # one-box A/B of the whole step (round 6, x2): 7.73 in-chain -> 8.68 deferred as per-layer launches -> 7.36 ms grouped
# Two-stream steps only: which stage boundaries of backward (1 = behind the last Downsample's backward, i.e. stage 4 done in that view,
# 2 = stages 4-3, 3 = stages 4-2) FORK the problems recorded so far onto an auxiliary stream as soon as BOTH views have passed them: the
# grouped launch then runs beside the remaining backward of the two views instead of behind it. Capped at DEFER_FORK_WGS workgroups (one
# per CU: the chains keep their LDS and wave slots). Three live streams (+ the communicator's under data parallelism): inside the four
# hardware queues (docs/experiments.md, round 3).
# DEFER_FORK_AT: the boundaries as decimal digits (2, 12, 123; 0 = no fork). MEASURED (round 6, one box, x2; no fork 7.50 ms):
# fork at 2 with 256 / 512 / 1024 / all workgroups 8.02 / 7.81 / 7.78 / 7.72; at 1: 7.68; at 3: 7.67; at 1,2: 7.96; at 1,2,3: 8.09 --
# whatever runs beside the two chains slows them by more than it hides (the finding of rounds 2-5 for every fat kernel). Off.
DEFER_FORK_AT = 0
DEFER_FORK_WGS = 256
DEFER_TWO_LANES = 1  # the short launches of the phase on the second view stream beside the two big ones: 7.40 -> 7.29 ms (x3); moving
#                      half of the wide problems over as well: equal (7.284 / 7.275)
DEFER_CHUNKS = 1     # pieces of the deferred phase when a gradient-ready hook is installed (parallel.GradReducer.install sets 3 under data parallelism)


class DeferredWgrads:
    def __init__(self):
        self.items, self.hooks, self.armed = [], [], False
        self.verify = None      # tests: a list that receives (dw, the same sum through the per-layer launches) for every layer of a flush
        self.crossed, self.events, self.aux = {}, {}, None
        self.calls = []

    def add(self, dout, x, dw, M, Nout, K, groups, in_scale, in_shift, act_in, extra=None):
        stage = self.crossed.get(torch.cuda.current_stream().cuda_stream, 0)      # boundaries this view's backward has passed
        self.items.append((stage, (dout, x, dw, M, Nout, K, groups, in_scale, in_shift, act_in, extra)))
        if not self.armed:
            self.armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def add_call(self, fn, tensors):
        """other work that only produces parameter gradients (the peak extractor's backward): runs on the phase's second lane"""
        self.calls.append((fn, tensors))
        if not self.armed:
            self.armed = True
            torch.autograd.Variable._execution_engine.queue_callback(self.flush)

    def _run_calls(self):
        calls, self.calls = self.calls, []
        for fn, tensors in calls:
            for t in tensors:
                t.record_stream(torch.cuda.current_stream())
            fn()

    def note_hook(self, params, n_before=0):
        """the block whose backward has just recorded the items [n_before:] reports its parameters when all of those have been issued"""
        stage = self.crossed.get(torch.cuda.current_stream().cuda_stream, 0)
        self.hooks.append((stage, params, {it[2].data_ptr() for _, it in self.items[n_before:]}))

    def boundary(self):
        """end of a Downsample's backward on the current stream (= in this view)"""
        if not self.armed:
            return
        st = torch.cuda.current_stream()
        c = self.crossed[st.cuda_stream] = self.crossed.get(st.cuda_stream, 0) + 1
        if str(c) not in str(int(DEFER_FORK_AT)) or not SIDE_STREAMS:
            return
        ev = torch.cuda.Event()
        ev.record(st)
        evs = self.events.setdefault(c, {})
        evs[st.cuda_stream] = ev
        if len(evs) < 2:
            return
        # both views are past boundary c: everything recorded in front of it goes out now, beside the rest of backward
        if self.aux is None:
            self.aux = torch.cuda.Stream(device=st.device)
            SIDE_STREAMS.append(self.aux)
        for e in evs.values():
            self.aux.wait_event(e)
        ready = [it for stage, it in self.items if stage < c]
        self.items = [(stage, it) for stage, it in self.items if stage >= c]
        hooks = [h for stage, h, _ in self.hooks if stage < c]
        self.hooks = [h3 for h3 in self.hooks if h3[0] >= c]
        with torch.cuda.stream(self.aux):
            self._issue(ready, DEFER_FORK_WGS)
            if GRAD_READY_HOOK is not None:
                for params in hooks:
                    GRAD_READY_HOOK(params)

    def _issue(self, items, max_wgs):
        if not items:
            return
        for it in items:                   # the launch reads tensors that were produced on the view streams
            it[0].record_stream(torch.cuda.current_stream())
            it[1].record_stream(torch.cuda.current_stream())
        ops.linear_bwd_weight_batch(items, max_wgs)
        if self.verify is not None:     # tests: the same problems through the per-layer launches, from the tensors as they are NOW
            tmp = {}
            for it in items:
                t = tmp.setdefault(it[2].data_ptr(), (it[2], torch.zeros_like(it[2])))[1]
                ops.wgrad_item(it[:2] + (t,) + it[3:])
            self.verify.extend((dw, t) for dw, t in tmp.values())

    def _issue_two_lanes(self, items):
        """The wide layers' problems (the 8-wave 128x128 classes: ~90 % of the flops, two launches, HBM-bound) on the current stream and
        everything else (five short launches that fill the chip badly: the C = 64 / 128 layers, Downsample, stem, the fp32 head — 0.35 ms
        of the 1.37 ms phase in an eager trace) on the idle view stream BESIDE them: no dependency chain on either side, so unlike a
        launch beside backward the co-resident kernels only fill each other's gaps. One fork edge, one join edge."""
        side = next((st for st in SIDE_STREAMS if st is not self.aux), None)
        heavy = [it for it in items if it[0].dtype == torch.bfloat16 and (len(it) <= 10 or it[10] is None) and it[3] % 128 == 0 and
                 it[4] % 128 == 0 and it[5] % 128 == 0]
        if not DEFER_TWO_LANES or side is None or not heavy or (len(heavy) == len(items) and not self.calls):
            self._issue(items, 0)
            self._run_calls()
            return
        light = [it for it in items if not any(it is h for h in heavy)]
        main = torch.cuda.current_stream()
        side.wait_stream(main)
        with torch.cuda.stream(side):
            self._issue(light, 0)
            self._run_calls()
        self._issue(heavy, 0)
        main.wait_stream(side)

    def flush(self):
        """runs on the thread that called backward(), once every backward node has been enqueued"""
        self.armed = False
        items, hooks, self.items, self.hooks = [it for _, it in self.items], [(h, ptrs) for _, h, ptrs in self.hooks], [], []
        self.crossed, self.events = {}, {}
        if not items and not hooks and not self.calls:
            return
        join_side_streams()
        chunks = max(1, int(DEFER_CHUNKS)) if GRAD_READY_HOOK is not None else 1
        if chunks == 1:
            self._issue_two_lanes(items)
            if GRAD_READY_HOOK is not None:
                for params, _ in hooks:
                    GRAD_READY_HOOK(params)
            return
        # Data parallelism: the phase goes out in `chunks` pieces of about equal gradient size, in backward order (late layers first:
        # the order the reducer cuts its buckets in), and every block reports its parameters as soon as its piece has been issued -- the
        # bucketed all-reduce of piece c runs on the communicator's stream beside the launches of piece c + 1.
        order, size = [], {}
        for it in items:
            ptr = it[2].data_ptr()
            if ptr not in size:
                order.append(ptr)
                size[ptr] = it[2].numel()
        total, acc, piece_of = float(sum(size.values())), 0.0, {}
        for ptr in order:
            piece_of[ptr] = min(chunks - 1, int(acc * chunks / total))
            acc += size[ptr]
        fired = [False] * len(hooks)
        for c in range(chunks):
            if c == 0:
                self._run_calls()          # (before any hook fires: a hook without recorded problems reports at the first piece)
            self._issue([it for it in items if piece_of[it[2].data_ptr()] == c], 0)
            for i, (params, ptrs) in enumerate(hooks):
                if not fired[i] and all(piece_of.get(p_, 0) <= c for p_ in ptrs):
                    fired[i] = True
                    GRAD_READY_HOOK(params)



"""Step runner: the whole training step (forward, backward, optimiser) captured ONCE into a HIP
graph and replayed -- the MI355X counterpart of the reference's static ``tf.Graph`` +
``sess.run`` (reference lib/models.py:267-351, :905-906).  Shapes are static there too
(``batch_size`` is baked into the placeholders, :272-282), which is what makes capture legal.

A step is ~190 short kernels; eager launches would be host-bound (>=3 us each), graph replay is
one submission.  With more than one rank the step is split at the gradient exchanges and the backward
pass runs in two phases so that the exchange of the early 96 % of the gradient bucket (decoder + dense
layers, final once the backward pass reaches the encoder convolutions) overlaps with the rest of it:

    graph A1 (fwd + backward phase 1)  ->  async all-reduce of flat_grad[:split]  (RCCL stream)
    graph A2 (backward phase 2: encoder convolutions, condition nets)  ||  ... all-reduce running
    async all-reduce of flat_grad[split:]  ->  wait both  ->  graph B (clip + update)

xGMI is point-to-point: between two GPUs the 65 MB bucket takes ~1.3 ms on one link against a 4 ms step, which
is what the overlap hides (phase 2 is ~1.3 ms).
"""
import os

import torch


# Stream-capture error mode of every graph this module records.  "thread_local": calls made by OTHER threads while the
# capture is open do not invalidate it -- with a process group alive, the collective backend's watchdog thread polls its
# events concurrently, and under the default "global" mode that intermittently killed the capture of the second backward
# graph (hipErrorStreamCaptureInvalidated from the next kernel launch).
_CAPTURE_MODE = "thread_local"


class GraphedTrainStep(object):
    def __init__(self, model, with_gan=False, grad_hook=None, use_graph=True, split=None):
        self.model, self.with_gan, self.grad_hook = model, with_gan, grad_hook
        # two-phase backward: on whenever gradients are exchanged (CAPE_DP_SPLIT=1 forces it for single-rank tests,
        # CAPE_DP_SPLIT=0 falls back to one sweep + one synchronous exchange)
        if split is None:
            env = os.environ.get("CAPE_DP_SPLIT", "")
            split = env == "1" or (grad_hook is not None and env != "0")
        self.split = bool(split) and not model.bug_compat
        model.split_backward = self.split
        # the exchange leaves the SUM of the ranks' gradients in the bucket; the optimiser kernels apply 1 / world (one bucket-sized
        # launch less per exchanged bucket, and clipping still sees the mean)
        # (kept on the RUNNER and handed to apply_updates explicitly -- it is baked by value into the captured update graph; step()
        # checks that the hook still is in the state this runner captured)
        self.grad_scale = 1.0
        if grad_hook is not None and hasattr(grad_hook, "defer_mean"):
            grad_hook.defer_mean = True
            self.grad_scale = grad_hook.grad_scale
        self.use_graph = use_graph          # (Adam's step count is a device counter advanced by the update kernel: capturable)
        B, d = model.batch_size, model.device
        M, Cn = model.input_num_verts, model.nn_input_channel
        z = lambda *s: torch.zeros(s, device=d, dtype=torch.float32)
        # network inputs live in row-padded buffers ([.., 3] views of [.., 4] rows): the first conv reads them with
        # aligned float4 loads directly instead of re-homing them every step
        zp = lambda: z(B, M, (Cn + 3) // 4 * 4)[:, :, :Cn]
        self.buf = dict(data_g=zp(), cond_g=z(B, model.cond_dim), cond2_g=z(B, model.cond2_dim), gt=z(B, M, Cn),
                        data_d=zp(), cond_d=z(B, model.cond_dim), cond2_d=z(B, model.cond2_dim),
                        eps=z(B, int(model.nz)))
        self.losses = {}
        self._gA = self._gA2 = self._gB = None

    def load_batch(self, **arrays):
        for k, v in arrays.items():
            self.buf[k].copy_(torch.as_tensor(v, dtype=torch.float32), non_blocking=True)

    # ---- the two halves of a step -------------------------------------------------------------------
    def _forward(self):
        m, b = self.model, self.buf
        if self.with_gan:
            return m.forward_losses(b['data_g'], b['cond_g'], b['cond2_g'], b['gt'], b['data_d'], b['cond_d'],
                                    b['cond2_d'], eps=b['eps'], reg_via_bucket=True)
        return m.forward_losses(b['data_g'], b['cond_g'], b['cond2_g'], b['gt'], eps=b['eps'], with_gan=False, reg_via_bucket=True)

    def _keep_losses(self, out):
        """The step's loss values stay where the loss kernels wrote them: under capture those tensors live in the graph's
        memory pool (kept alive by these references, so nothing later in the graph reuses them) and every replay rewrites
        them in place -- no copy launches."""
        for k in ('loss_g', 'loss_d', 'recon', 'latent', 'edge'):
            if k in out and torch.is_tensor(out[k]):
                self.losses[k] = out[k].detach().reshape(())

    def _fwd_bwd(self):
        """forward + the whole backward pass (single-rank path; also what bench.py's per-launch timing replays)."""
        out = self._forward()
        self.model.backward_to_flat(out)
        self._keep_losses(out)

    def _fwd_bwd1(self):
        out = self._forward()
        self.model.backward_phase1(out)
        self._keep_losses(out)

    def _bwd2(self):
        self.model.backward_phase2()

    def _groups(self):
        return ('g', 'd') if self.with_gan else ('g',)

    def _update(self):
        for grp in self._groups():
            self.model.apply_updates(grp, grad_scale=self.grad_scale)

    def _exchange(self):
        """Synchronous exchange of every bucket (unsplit path)."""
        if self.grad_hook is not None:
            for grp in self._groups():
                if not (grp == 'd' and self.model.bug_compat):
                    self.grad_hook(self.model._opt_state[grp]['flat_grad'])

    def _start(self, t):
        h = self.grad_hook
        if h is None or t.numel() == 0:
            return None
        if hasattr(h, 'start'):
            return h.start(t)
        h(t)
        return None

    def _finish(self, works):
        for w in works:
            if w is not None:
                self.grad_hook.finish(w)

    def _exchange_early(self):
        st = self.model._opt_state
        works = [self._start(st['g']['flat_grad'][:st['g']['split_off']])]
        if self.with_gan:
            works.append(self._start(st['d']['flat_grad']))
        return works

    def _exchange_late(self):
        st = self.model._opt_state['g']
        return [self._start(st['flat_grad'][st['split_off']:])]

    def _split_step_eager(self):
        self._fwd_bwd1()
        works = self._exchange_early()
        self._bwd2()
        works += self._exchange_late()
        self._finish(works)
        self._update()

    # ---- capture / replay --------------------------------------------------------------------------
    def capture(self, warmup=2, preserve_state=False):
        """Capture the step into HIP graphs (after ``warmup`` eager passes that load every kernel and settle the
        allocator).  The warm-up passes run real updates (on whatever batch the input buffers hold, with the current
        learning-rate scalars): ``preserve_state`` snapshots variables and optimiser buffers first and restores them
        afterwards, so that a training run's trajectory is exactly that of the un-captured step (fit())."""
        if not self.use_graph:
            return self
        saved = None
        if preserve_state and warmup > 0:
            saved = {grp: {k: st[k].detach().clone() for k in ('flat', 'm', 'v', 't') if k in st}
                     for grp, st in self.model._opt_state.items()}
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):        # allocator warm-up outside capture
                if self.split:
                    self._split_step_eager()
                else:
                    self._fwd_bwd()
                    self._exchange()
                    self._update()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        if saved is not None:
            with torch.no_grad():
                for grp, bufs in saved.items():
                    for k, v in bufs.items():
                        self.model._opt_state[grp][k].copy_(v)
            torch.cuda.synchronize()
        self._gA = torch.cuda.CUDAGraph()
        if self.split:
            with torch.cuda.graph(self._gA, capture_error_mode=_CAPTURE_MODE):
                self._fwd_bwd1()
            self._gA2 = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._gA2, pool=self._gA.pool(), capture_error_mode=_CAPTURE_MODE):
                self._bwd2()
            self._gB = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._gB, pool=self._gA.pool(), capture_error_mode=_CAPTURE_MODE):
                self._update()
            return self
        exchange = self.grad_hook is not None
        with torch.cuda.graph(self._gA, capture_error_mode=_CAPTURE_MODE):
            self._fwd_bwd()
            if not exchange:
                self._update()
        if exchange:
            self._gB = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._gB, pool=self._gA.pool(), capture_error_mode=_CAPTURE_MODE):
                self._update()
        return self

    def step(self):
        m = self.model
        h = self.grad_hook
        if h is not None and hasattr(h, "defer_mean"):
            # the captured update divides by what the hook left undone at capture time: both must still agree
            if not h.defer_mean or abs(h.grad_scale - self.grad_scale) > 0:
                raise RuntimeError("the gradient hook's state changed since this runner was built (defer_mean / world size): the "
                                   "captured update would scale the gradients by %g where the hook now expects %g" % (self.grad_scale, h.grad_scale))
        m.set_learning_rates(self._groups())
        if self._gA is None:
            if self.split:
                self._split_step_eager()
            else:
                self._fwd_bwd()
                self._exchange()
                self._update()
        elif self.split:
            self._gA.replay()
            works = self._exchange_early()
            self._gA2.replay()
            works += self._exchange_late()
            self._finish(works)
            self._gB.replay()
        else:
            self._gA.replay()
            if self._gB is not None:
                self._exchange()
                self._gB.replay()
        # a replay rewrites the variables without Python seeing it (no version bump, no apply_updates call): the piece planes
        # of the fp16 two-piece contractions, refreshed at the START of the captured step, are one update behind afterwards --
        # the next pass outside the graph (validation inside fit(), predict, encode) must rebuild them
        m._pieces_dirty = True
        m.global_step += len(self._groups())

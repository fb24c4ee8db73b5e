"""Step runner: the whole training step (forward, backward, optimiser) captured ONCE into a HIP
graph and replayed -- the MI355X counterpart of the reference's static ``tf.Graph`` +
``sess.run`` (reference lib/models.py:267-351, :905-906).  Shapes are static there too
(``batch_size`` is baked into the placeholders, :272-282), which is what makes capture legal.

A step is ~190 short kernels; eager launches would be host-bound (>=3 us each), graph replay is
one submission.  With more than one rank the step is split at the gradient exchange:
graph A (fwd + bwd -> flat gradients), eager RCCL all-reduce, graph B (clip + update).
"""
import torch


class GraphedTrainStep(object):
    def __init__(self, model, with_gan=False, grad_hook=None, use_graph=True):
        self.model, self.with_gan, self.grad_hook = model, with_gan, grad_hook
        self.use_graph = use_graph and model.optimizer != "adam"   # Adam keeps a host-side step counter
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
        self._gA = self._gB = None

    def load_batch(self, **arrays):
        for k, v in arrays.items():
            self.buf[k].copy_(torch.as_tensor(v, dtype=torch.float32), non_blocking=True)

    # ---- the two halves of a step -------------------------------------------------------------------
    def _fwd_bwd(self):
        m, b = self.model, self.buf
        if self.with_gan:
            out = m.forward_losses(b['data_g'], b['cond_g'], b['cond2_g'], b['gt'], b['data_d'], b['cond_d'],
                                   b['cond2_d'], eps=b['eps'], reg_via_bucket=True)
        else:
            out = m.forward_losses(b['data_g'], b['cond_g'], b['cond2_g'], b['gt'], eps=b['eps'], with_gan=False, reg_via_bucket=True)
        m.backward_to_flat(out)
        dst, src = [], []
        for k in ('loss_g', 'loss_d', 'recon', 'latent', 'edge'):
            if k in out and torch.is_tensor(out[k]):
                if k not in self.losses:
                    self.losses[k] = torch.zeros((), device=m.device)
                dst.append(self.losses[k])
                src.append(out[k].detach().reshape(()))
        if dst:
            torch._foreach_copy_(dst, src)

    def _groups(self):
        return ('g', 'd') if self.with_gan else ('g',)

    def _update(self):
        for grp in self._groups():
            self.model.apply_updates(grp)

    def _exchange(self):
        if self.grad_hook is not None:
            for grp in self._groups():
                if not (grp == 'd' and self.model.bug_compat):
                    self.grad_hook(self.model._opt_state[grp]['flat_grad'])

    # ---- capture / replay --------------------------------------------------------------------------
    def capture(self, warmup=2):
        if not self.use_graph:
            return self
        s = torch.cuda.Stream()
        s.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s):
            for _ in range(warmup):        # allocator warm-up outside capture
                self._fwd_bwd()
                self._exchange()
                self._update()
        torch.cuda.current_stream().wait_stream(s)
        torch.cuda.synchronize()
        split = self.grad_hook is not None
        self._gA = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._gA):
            self._fwd_bwd()
            if not split:
                self._update()
        if split:
            self._gB = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._gB, pool=self._gA.pool()):
                self._update()
        return self

    def step(self):
        m = self.model
        m.set_learning_rates()
        if self._gA is None:
            self._fwd_bwd()
            self._exchange()
            self._update()
        else:
            self._gA.replay()
            if self._gB is not None:
                self._exchange()
                self._gB.replay()
        m.global_step += len(self._groups())

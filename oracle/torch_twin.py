"""Torch-CPU autograd twin of oracle/cape_oracle.py -- gradient oracle.  TEST INFRASTRUCTURE ONLY.

Same op order, layouts and variable names as the numpy restatement (and therefore as the
reference lib/models.py lines cited there); exists because the reference obtains gradients from
``tf.gradients`` (lib/models.py:460,465) and numpy has no autodiff.  Its forward is checked
against the numpy oracle in tests/test_oracle.py; its backward is what the HIP gradient
kernels are compared with.  Runs in float64 by default.
"""
import numpy as np
import scipy.sparse as sp
import torch

from . import cape_oracle as co


def _sparse(mat, dtype):
    m = sp.coo_matrix(mat)
    idx = torch.from_numpy(np.vstack([m.row, m.col]).astype(np.int64))
    return torch.sparse_coo_tensor(idx, torch.from_numpy(m.data.astype(np.float64)).to(dtype), m.shape).coalesce()


def chebyshev5(x, L, W, K):
    """lib/models.py:69-103 in torch (same [M, Fin*N] layout path)."""
    N, M, Fin = x.shape
    Lr = _sparse(co.rescale_L(sp.csr_matrix(L), 2), x.dtype)
    x0 = x.permute(1, 2, 0).reshape(M, Fin * N)
    stack = [x0]
    if K > 1:
        x1 = torch.sparse.mm(Lr, x0)
        stack.append(x1)
    for _ in range(2, K):
        x2 = 2 * torch.sparse.mm(Lr, x1) - x0
        stack.append(x2)
        x0, x1 = x1, x2
    xs = torch.stack(stack, 0).reshape(K, M, Fin, N).permute(3, 1, 2, 0).reshape(N * M, Fin * K)
    return (xs @ W).reshape(N, M, W.shape[1])


def poolwT(x, P):
    """lib/models.py:129-152."""
    N, M, Fin = x.shape
    Pm = _sparse(P, x.dtype)
    xt = x.permute(1, 2, 0).reshape(M, Fin * N)
    y = torch.sparse.mm(Pm, xt).reshape(P.shape[0], Fin, N)
    return y.permute(2, 0, 1).contiguous()


def bias_act(x, b, kind):
    z = x + b
    if kind == "b1leakyrelu":
        return torch.nn.functional.leaky_relu(z, 0.2)
    if kind == "b1tanh":
        return torch.tanh(z)
    return torch.relu(z)


def forced_act(z, slope, sign, rows=None):
    """(leaky-)ReLU whose branch per unit is GIVEN (``sign``: bool, True = positive side) instead of decided by z > 0.
    Used by the gradient-parity tests: the device's fp32 forward and this fp64 graph disagree about the branch of the few
    units whose pre-activation lies within fp32 rounding of zero; with the device's pattern imposed both graphs are the same
    piecewise-linear function and the gradients compare at arithmetic accuracy.  ``rows``: the pattern covers only these
    vertex rows of z (a layer evaluated on the kept vertices of the following row-selection pool); the other rows, whose
    gradient the pool discards, keep their own sign."""
    sign = torch.as_tensor(sign, dtype=torch.bool)
    if rows is not None:
        full = (z.detach() > 0)
        full[:, torch.as_tensor(np.asarray(rows), dtype=torch.long)] = sign
        sign = full
    assert tuple(sign.shape) == tuple(z.shape), (tuple(sign.shape), tuple(z.shape))
    return torch.where(sign, z, slope * z)


def group_norm(x, gamma, beta, G=32, eps=1e-5):
    """lib/models.py:693-709."""
    xt = x.permute(0, 2, 1)
    N, C, V = xt.shape
    G = min(G, C)
    xg = xt.reshape(-1, G, C // G, V)          # free leading dimension, as the reference writes it (see cape_oracle.group_norm)
    mean = xg.mean(dim=(2, 3), keepdim=True)
    var = ((xg - mean) ** 2).mean(dim=(2, 3), keepdim=True)
    xg = (xg - mean) / torch.sqrt(var + eps)
    out = xg.reshape(-1, C, V) * gamma.reshape(1, C, 1) + beta.reshape(1, C, 1)
    return out.permute(0, 2, 1).contiguous()


def fit_cond_dim(x, y):
    return y.reshape(x.shape[0], 1, -1).expand(x.shape[0], x.shape[1], y.shape[-1])


def edge_loss_calc(pred, gt, vpe):
    vpe = torch.as_tensor(np.asarray(vpe), dtype=torch.long)
    ev = lambda v: v[:, vpe[:, 0], :] - v[:, vpe[:, 1], :]
    d = ev(pred) - ev(gt)
    return torch.sqrt((d * d).sum(-1)).mean()


class TwinCAPE(co.OracleCAPE):
    """OracleCAPE with torch tensors; variables become leaf tensors with requires_grad."""

    def __init__(self, *a, **kw):
        tdtype = kw.pop("tdtype", torch.float64)
        super(TwinCAPE, self).__init__(*a, **kw)
        self.td = tdtype
        self.params = {}
        self.forced_signs = None      # collections.deque of recorded branch patterns, consumed in execution order
        self.flip_log = []            # per site: number of units whose own sign differs from the imposed one
        self.sign_log = None          # a list -> plain evaluation records (own pattern, pool matrix or None) per site
        self.forced_l1_sign = None    # sign(pred - gt) recorded by the device: |d| is then evaluated as sign * d

    def _site(self, z, slope, plain, pool=None):
        """One (leaky-)ReLU site: ``plain(z)`` unless a recorded branch pattern is being replayed."""
        if self.forced_signs is None:
            if self.sign_log is not None:
                self.sign_log.append(((z.detach() > 0), pool))
            return plain(z) if plain is not None else torch.where(z > 0, z, slope * z)
        sign = torch.as_tensor(self.forced_signs.popleft(), dtype=torch.bool)
        rows = None
        if sign.dim() == 3 and z.dim() == 3 and sign.shape[1] != z.shape[1]:
            P = sp.csr_matrix(pool)                      # row selection: output row r = input row indices[r]
            assert P.shape[0] == sign.shape[1] and P.nnz == P.shape[0], "pattern rows do not match the pool"
            rows = P.indices
        own = (z.detach() > 0) if rows is None else (z.detach()[:, torch.as_tensor(rows, dtype=torch.long)] > 0)
        self.flip_log.append(int((own != sign).sum()))
        return forced_act(z, slope, sign, rows)

    def _p(self, name_arr):
        full, arr = name_arr
        if full not in self.params:
            self.params[full] = torch.tensor(arr, dtype=self.td, requires_grad=True)
        return self.params[full]

    def _get(self, name, shape, kind, tag, **kw):
        arr = self.vs.get(name, shape, kind, tag, **kw)
        return self._p((self.vs.full(name), arr))

    def _weight(self, shape):
        return self._get('weights', shape, 'trunc_normal', 'conv', stddev=0.1)

    def _bias(self, shape):
        return self._get('bias', shape, 'const', 'bias', value=0.1)

    def _dense(self, x, units, activation=None):
        with self.vs.scope('dense'):
            k = self._get('kernel', (x.shape[-1], units), 'glorot_uniform', 'fc_kernel')
            b = self._get('bias', (units,), 'zeros', 'fc_bias')
        y = x @ k + b
        if activation == 'leaky_relu':
            y = self._site(y, 0.2, lambda z: torch.nn.functional.leaky_relu(z, 0.2))
        return y

    def filter(self, x, L, Fout, K):
        return chebyshev5(x, L, self._weight((x.shape[-1] * K, Fout)), K)

    def brelu(self, x, pool=None):
        if self.activation == 'b2relu':
            b = self._bias((1, x.shape[1], x.shape[2]))
        else:
            b = self._bias((1, 1, x.shape[2]))
        if (self.forced_signs is None and self.sign_log is None) or self.activation == 'b1tanh':
            return bias_act(x, b, self.activation)
        return self._site(x + b, 0.2 if self.activation == 'b1leakyrelu' else 0.0, None, pool=pool)

    def _t(self, a):
        return a if torch.is_tensor(a) else torch.as_tensor(np.asarray(a), dtype=self.td)

    def cnp(self, x, i, name):
        with self.vs.scope(name):
            x = self.filter(x, self.Laplacian[i], self.out_channels[i], self.poly_order[i])
            return poolwT(self.brelu(x, pool=self.Downsample_mtx[i]), self.Downsample_mtx[i])

    def udn(self, x, out_channels, i, name):
        with self.vs.scope(name):
            x = poolwT(x, self.Upsample_mtx[-i - 1])
            x = self.filter(x, self.Laplacian[-i - 2], out_channels[-i - 1], self.poly_order[-i - 1])
            return self.brelu(x)

    def cnp_d(self, x, i, name):
        with self.vs.scope(name):
            x = self.filter(x, self.Laplacian_d[i], self.out_channels[i], self.poly_order_d[i])
            return poolwT(self.brelu(x, pool=self.Downsample_mtx_d[i]), self.Downsample_mtx_d[i])

    def gn(self, x, name):
        with self.vs.scope(name):
            C = x.shape[-1]
            gamma = self._get('gamma', (C,), 'ones', 'gn')
            beta = self._get('beta', (C,), 'zeros', 'gn')
        return group_norm(x, gamma, beta)

    def res_block(self, x_in, i, name):
        with self.vs.scope(name):
            with self.vs.scope('filter_1'):
                x1 = self.filter(x_in, self.Laplacian[i], self.out_channels[i], self.poly_order[i])
            with self.vs.scope('bias_relu_1'):
                x1 = self.brelu(x1)
            with self.vs.scope('filter_2'):
                x2 = self.filter(x1, self.Laplacian[i], self.out_channels[i], self.poly_order[i])
            if x_in.shape[-1] != x2.shape[-1]:
                with self.vs.scope('1x1-conv'):
                    x_in = self.filter(x_in, self.Laplacian[i], x2.shape[-1], 1)
            x2 = x2 + x_in
            with self.vs.scope('bias_relu_2'):
                x2 = self.brelu(x2)
            return poolwT(x2, self.Downsample_mtx[i])

    def res_block_decoder(self, x_in, i, name):
        Fi, Lm = self.out_channels[-i - 1], self.Laplacian[-i - 2]
        with self.vs.scope(name):
            xu = poolwT(x_in, self.Upsample_mtx[-i - 1])
            x = self._site(self.gn(xu, 'group_norm'), 0.0, torch.relu)
            with self.vs.scope('graph_linear_1'):
                x = self.filter(x, Lm, Fi // 2, 1)
            x = self._site(self.gn(x, 'group_norm_1'), 0.0, torch.relu)
            with self.vs.scope('graph_conv'):
                x = self.filter(x, Lm, Fi // 2, self.poly_order[-i - 1])
            x = self._site(self.gn(x, 'group_norm_2'), 0.0, torch.relu)
            with self.vs.scope('graph_linear_2'):
                x = self.filter(x, Lm, Fi, 1)
            if xu.shape[-1] != x.shape[-1]:
                with self.vs.scope('graph_linear_input'):
                    xu = self.filter(xu, Lm, x.shape[-1], 1)
            return x + xu

    def res_block_affine(self, x, i, name):
        Lm = self.Laplacian[-i - 2]
        with self.vs.scope(name):
            x = poolwT(x, self.Upsample_mtx[-i - 1])
            with self.vs.scope('graph_conv'):
                x_gc = self._site(self.filter(x, Lm, self.out_channels[-i - 1] // 2, self.poly_order[-i - 1]), 0.0, torch.relu)
            with self.vs.scope('affine'):
                x_aff = self.filter(x, Lm, x_gc.shape[-1], 1)
            return x_aff + x_gc

    def condition(self, y, name, nz_cond, nlayers=1):
        y = self._t(y)
        y_dim = y.shape[-1]
        with self.vs.scope('condition_{}'.format(name)):
            if nlayers == 1:
                with self.vs.scope('fc1'):
                    y = self._dense(y, nz_cond)
            else:
                if nz_cond < y_dim // 2:
                    n1 = y_dim // 2
                elif nz_cond < y_dim * 2:
                    n1 = y_dim
                else:
                    n1 = nz_cond // 2
                with self.vs.scope('fc1'):
                    y = self._dense(y, n1, 'leaky_relu')
                with self.vs.scope('fc2'):
                    y = self._dense(y, nz_cond)
        return y

    def encoder(self, x, y, y2):
        x = self._t(x)
        if self.cond_encoder:
            x = torch.cat([x, fit_cond_dim(x, y), fit_cond_dim(x, y2)], -1)
        with self.vs.scope('generator'), self.vs.scope('encoder'):
            for i in range(len(self.out_channels)):
                if self.use_res_block:
                    x = self.res_block(x, i, 'encoder_resblock{}'.format(i + 1))
                else:
                    x = self.cnp(x, i, 'encoder_conv{}'.format(i + 1))
            if self.reduce_dim > 0:
                with self.vs.scope('1x1-conv'):
                    x = self.filter(x, self.Laplacian[-1], self.out_channels[-1] // self.reduce_rate, 1)
            x = x.reshape(x.shape[0], -1)
            with self.vs.scope('fc_mean'):
                z_mean = self._dense(x, int(self.nz))
            with self.vs.scope('fc_var'):
                z_var = self._dense(x, int(self.nz))
        return z_mean, z_var

    def decoder_cond_vert(self, x, y, y2):
        x, y, y2 = self._t(x), self._t(y), self._t(y2)
        N = x.shape[0]
        with self.vs.scope('generator'), self.vs.scope('decoder'):
            with self.vs.scope('fc1'):
                out_nodes = int(self.p[-1] * self.out_channels[-1]) // self.reduce_rate
                x = self._dense(x, out_nodes, 'leaky_relu')
            x = x.reshape(N, int(self.p[-1]), -1)
            if self.reduce_dim > 0:
                with self.vs.scope('1x1-conv'):
                    x = self.filter(x, self.Laplacian[-1], self.out_channels[-1], 1)
            x = torch.cat([x, fit_cond_dim(x, y), fit_cond_dim(x, y2)], -1)
            for i in range(len(self.out_channels)):
                if self.use_res_block_dec:
                    if not self.affine:
                        x = self.res_block_decoder(x, i, 'decoder_resblock_cmr{}'.format(i + 1))
                    else:
                        x = self.res_block_affine(x, i, 'decoder_resblock_affine{}'.format(i + 1))
                else:
                    x = self.udn(x, self.out_channels, i, 'decoder_conv{}'.format(i + 1))
                x = torch.cat([x, fit_cond_dim(x, y), fit_cond_dim(x, y2)], -1)
            with self.vs.scope('outputs'):
                x = self.filter(x, self.Laplacian[0], int(self.nn_input_channel), self.poly_order[0])
                x = x + self._bias((1, x.shape[1], x.shape[2]))
        return x

    def vae_sampling(self, z_mean, z_logvar, eps):
        return z_mean + torch.sqrt(torch.exp(z_logvar)) * self._t(eps)

    def generator(self, x, y, y2, eps):
        z_mean, z_logvar = self.encoder(x, y, y2)
        z = self.vae_sampling(z_mean, z_logvar, eps)
        x_hat = self.decoder_cond_vert(torch.cat([z, y, y2], 1), y, y2)
        return x_hat, z_mean, z_logvar

    def discriminator(self, x, y, y2):
        x = self._t(x)
        x = torch.cat([x, fit_cond_dim(x, y), fit_cond_dim(x, y2)], -1)
        with self.vs.scope('discriminator'):
            with self.vs.scope('shared'):
                for i in range(len(self.Downsample_mtx_d)):
                    x = self.cnp_d(x, i, 'conv{}'.format(i + 1))
            with self.vs.scope('prediction_map'):
                return self.filter(x, self.Laplacian_d[-1], 1, self.poly_order[-1])

    def losses(self, g_out, g_gt, z_mean, z_logvar, d_real=None, d_fake=None, smooth=0.1):
        g_gt = self._t(g_gt)
        out = {}
        diff = g_out - g_gt
        if self.which_loss == 'l1':
            if self.forced_l1_sign is not None:
                sg = torch.as_tensor(np.asarray(self.forced_l1_sign), dtype=diff.dtype)
                self.flip_log.append(int((sg != torch.sign(diff.detach())).sum()))
                out['recon'] = (sg * diff).mean()
            else:
                out['recon'] = diff.abs().mean()
        elif self.which_loss == 'huber':
            a = diff.abs()
            out['recon'] = torch.where(a <= 0.1, 0.5 * a * a, 0.1 * a - 0.005).mean()
        else:
            out['recon'] = (diff * diff).mean()
        out['latent'] = (-0.5 * (1 + z_logvar - z_mean ** 2 - torch.exp(z_logvar)).sum(1)).mean()
        vr = self._t(self.verts_ref)
        out['edge'] = edge_loss_calc(g_out + vr, g_gt + vr, self.vpe)
        reg = sum(0.5 * (self.params[n] ** 2).sum() for n in self.params
                  if self.vs.kinds[n] == 'fc_kernel' and n.startswith('generator'))
        out['fc_reg_g'] = self.regularization * self.regularization * reg
        total = out['recon'] * self.lambda_l1 + out['edge'] * self.lambda_edge + \
            out['latent'] * self.lambda_latent + out['fc_reg_g']
        bce = torch.nn.functional.binary_cross_entropy_with_logits
        if d_fake is not None:
            out['gan_g'] = bce(d_fake, torch.full_like(d_fake, 1 - smooth))
            total = total + out['gan_g'] * self.lambda_gan
            if d_real is not None:
                out['gan_d'] = bce(d_real, torch.full_like(d_real, 1 - smooth)) + \
                    bce(d_fake, torch.full_like(d_fake, smooth))
                out['loss_d'] = out['gan_d'] * self.lambda_gan
        out['loss_g'] = total
        return out

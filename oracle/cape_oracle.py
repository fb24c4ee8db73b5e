"""CPU oracle: numpy/scipy restatement of the reference's Chebyshev mesh-conv hot path.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  Every function cites the reference
``file:line`` it follows (paths relative to the reference checkout).  The arithmetic of
the reference lives in TensorFlow 1.13.2 kernels (``SparseTensorDenseMatMul``, ``MatMul``,
``tf.layers.dense``, ``tf.nn.*``) which are NOT vendored and cannot be installed here;
what is restated is the reference's *Python graph assembly* (op order, layouts, indices,
variable names) with the documented semantics of those TF ops.

PARITY PINNING: the reference ships no tests and no golden vectors.  This
oracle is pinned by ``oracle/make_golden.py``, which executes the reference's own
``lib/models.py`` graph-assembly code on a numpy implementation of the TF1 ops it calls
(``oracle/tf1_numpy_shim``) and stores inputs/outputs under ``tests/golden``; the
restatement below must reproduce those (tests/test_oracle_golden.py).  TF1's own C++
kernels remain unpinned ("parity unpinned" at that level; stated in DESIGN.md).

Tiers: ``dtype=np.float64`` = truth; ``dtype=np.float32`` = stand-in for the
TF1 CPU path in the reference's op order (scipy CSR@dense accumulates each row
sequentially over column-sorted entries, like TF's CPU SparseTensorDenseMatMul).
"""
import numpy as np
import scipy.sparse as sp

from . import weights as winit


# --------------------------------------------------------------------------------------
# operator precompute (lib/mesh_sampling.py:10-38)
# --------------------------------------------------------------------------------------
def laplacian(W):
    """lib/mesh_sampling.py:10-29, normalized branch: I - D^-1/2 W D^-1/2."""
    d = W.sum(axis=0)
    d = d + np.spacing(np.array(0, W.dtype))
    d = 1 / np.sqrt(d)
    Dm = sp.diags(np.asarray(d).squeeze(), 0)
    eye = sp.identity(d.size, dtype=W.dtype)
    return (eye - Dm * W * Dm).tocsr()


def rescale_L(L, lmax=2):
    """lib/mesh_sampling.py:31-38: L/(lmax/2) - I."""
    M = L.shape[0]
    eye = sp.identity(M, format="csr", dtype=L.dtype)
    L = L / (lmax / 2)
    return (L - eye).tocsr()


# --------------------------------------------------------------------------------------
# graph operators (lib/models.py:69-152)
# --------------------------------------------------------------------------------------
def chebyshev5(x, L, W, K):
    """lib/models.py:69-103.  x [N,M,Fin]; L scipy sparse (un-rescaled Laplacian);
    W [Fin*K, Fout] with row index fin*K+k (:97-101).  Same op order as the reference,
    including the two layout shuffles (:81-83, :97-99)."""
    dt = x.dtype
    N, M, Fin = x.shape
    Lr = rescale_L(sp.csr_matrix(L), lmax=2)            # :74-75
    Lr = sp.csr_matrix(Lr, dtype=dt)
    Lr.sort_indices()                                   # tf.sparse_reorder :79
    x0 = np.transpose(x, (1, 2, 0)).reshape(M, Fin * N)  # :81-82
    stack = [x0]
    if K > 1:
        x1 = Lr @ x0                                    # :91
        stack.append(x1)
    for _ in range(2, K):
        x2 = 2 * (Lr @ x1) - x0                         # :94
        stack.append(x2)
        x0, x1 = x1, x2
    xs = np.stack(stack, axis=0)                        # K x M x Fin*N
    xs = xs.reshape(K, M, Fin, N)                       # :97
    xs = np.transpose(xs, (3, 1, 2, 0))                 # :98  N x M x Fin x K
    xs = np.ascontiguousarray(xs).reshape(N * M, Fin * K)  # :99
    y = xs @ W.astype(dt)                               # :102
    return y.reshape(N, M, W.shape[1])


def poolwT(x, P):
    """lib/models.py:129-152: y[n] = P @ x[n] through the [M, Fin*N] layout."""
    dt = x.dtype
    N, M, Fin = x.shape
    Mp = P.shape[0]
    Pm = sp.csr_matrix(P, dtype=dt)
    Pm.sort_indices()
    xt = np.transpose(x, (1, 2, 0)).reshape(M, Fin * N)   # :147-148
    y = Pm @ xt                                           # :149
    y = y.reshape(Mp, Fin, N)                             # :150
    return np.ascontiguousarray(np.transpose(y, (2, 0, 1)))  # :151


def leaky_relu(x, alpha=0.2):
    """tf.nn.leaky_relu default alpha=0.2 (lib/models.py:109)."""
    return np.maximum(x, alpha * x) if alpha <= 1 else np.where(x > 0, x, alpha * x)


def relu(x):
    return np.maximum(x, 0)


def bias_act(x, b, kind):
    """lib/models.py:105-127: b1leakyrelu / b1tanh / b1relu (bias [1,1,F]), b2relu ([1,M,F])."""
    z = x + b.astype(x.dtype)
    if kind == "b1leakyrelu":
        return leaky_relu(z)
    if kind == "b1tanh":
        return np.tanh(z)
    if kind in ("b1relu", "b2relu"):
        return relu(z)
    raise ValueError(kind)


def group_norm(x, gamma, beta, G=32, eps=1e-5):
    """lib/models.py:693-709: [N,V,C] -> [N,C,V] -> [N,G,C/G,V], population moments over
    axes (2,3), per-channel gamma/beta, back to [N,V,C]."""
    dt = x.dtype
    xt = np.transpose(x, (0, 2, 1))
    N, C, V = xt.shape
    G = min(G, C)
    # the reference reshapes with a free leading dimension (tf.reshape(x, [-1, G, C // G, V]), :698): when G does not
    # divide C the rows of the [N*C, V] matrix are still taken C // G at a time (C = 48: one channel row per group, i.e.
    # a per-(sample, channel) normalisation) -- restated as written
    xg = xt.reshape(-1, G, C // G, V)
    mean = xg.mean(axis=(2, 3), keepdims=True)
    var = ((xg - mean) ** 2).mean(axis=(2, 3), keepdims=True)
    xg = (xg - mean) / np.sqrt(var + dt.type(eps))
    out = xg.reshape(-1, C, V) * gamma.astype(dt).reshape(1, C, 1) + beta.astype(dt).reshape(1, C, 1)
    return np.ascontiguousarray(np.transpose(out, (0, 2, 1)))


def fit_cond_dim(x, y):
    """lib/models.py:813-832: tile [N,C] to [N,M,C]."""
    N, M = x.shape[0], x.shape[1]
    return y.reshape(N, 1, y.shape[-1]) * np.ones((N, M, y.shape[-1]), dtype=x.dtype)


def dense(x, kernel, bias, activation=None):
    """tf.layers.dense: x @ kernel + bias (lib/models.py:496,506,510,557,560,582)."""
    y = x @ kernel.astype(x.dtype) + bias.astype(x.dtype)
    if activation == "leaky_relu":
        y = leaky_relu(y)
    return y


def edge_loss_calc(pred, gt, vpe):
    """lib/losses.py:9-25: mean Euclidean norm of (edge vectors of pred - edge vectors of gt)."""
    ev = lambda v: v[:, vpe[:, 0], :] - v[:, vpe[:, 1], :]
    diff = ev(pred) - ev(gt)
    return np.sqrt((diff * diff).sum(-1)).mean()


def sigmoid_xent(logits, labels):
    """tf.nn.sigmoid_cross_entropy_with_logits: max(x,0) - x*z + log(1+exp(-|x|))."""
    return np.maximum(logits, 0) - logits * labels + np.log1p(np.exp(-np.abs(logits)))


# --------------------------------------------------------------------------------------
# variable store keyed by the TF variable names of SURVEY appendix B
# --------------------------------------------------------------------------------------
class VarStore(object):
    def __init__(self, seed=123):
        self.seed = seed
        self.vars = {}          # name -> float32 array
        self.kinds = {}         # name -> 'conv' | 'bias' | 'fc_kernel' | 'fc_bias' | 'gn'
        self._scope = []

    class _Scope(object):
        def __init__(self, store, name):
            self.store, self.name = store, name

        def __enter__(self):
            self.store._scope.append(self.name)

        def __exit__(self, *a):
            self.store._scope.pop()

    def scope(self, name):
        return VarStore._Scope(self, name)

    def full(self, name):
        return "/".join(self._scope + [name])

    def get(self, name, shape, kind, tag, **kw):
        full = self.full(name)
        if full not in self.vars:
            self.vars[full] = winit.init_variable(kind, shape, self.seed, full, **kw)
            self.kinds[full] = tag
        v = self.vars[full]
        assert tuple(v.shape) == tuple(int(s) for s in shape), (full, v.shape, shape)
        return v


class OracleCAPE(object):
    """Restatement of lib/models.py classes base_model (:13-227) + CAPE (:230-832):
    forward networks and losses.  Constructor mirrors :15-21 / :235-238."""

    def __init__(self, L, D, U, L_d, D_d, lr_scaler=0.1, lambda_gan=0.1, use_res_block=False,
                 use_res_block_dec=True, nz_cond2=8, cond2_dim=4, Kd=3, n_layer_cond=1,
                 cond_encoder=True, reduce_dim=True, affine=False, lr_warmup=False,
                 optim_condnet=True, F=None, K=None, p=None, nz=18, loss='l1', nn_input_channel=3,
                 filter='chebyshev5', activation='b1leakyrelu', pool='poolwT', unpool='poolwT',
                 cond_dim=0, nz_cond=0, regularization=0, batch_size=32, seed=123,
                 lambda_recon=1.0, lambda_edge=0.0, lambda_latent=1e-3, dtype=np.float64,
                 verts_ref=None, vpe=None, **unused):
        self.Laplacian, self.Downsample_mtx, self.Upsample_mtx, self.p = L, D, U, p
        self.Laplacian_d, self.Downsample_mtx_d = L_d, D_d
        self.out_channels, self.poly_order = F, K
        self.poly_order_d = [Kd] * len(F)                       # :241
        self.use_res_block, self.use_res_block_dec = use_res_block, use_res_block_dec
        self.nz, self.nz_cond, self.nz_cond2 = nz, nz_cond, nz_cond2
        self.cond_dim, self.cond2_dim, self.n_layer_cond = cond_dim, cond2_dim, n_layer_cond
        self.cond_encoder, self.affine = cond_encoder, affine
        self.reduce_dim = reduce_dim
        if self.reduce_dim > 0:                                 # :254-259
            self.reduce_rate = F[-1] // self.reduce_dim
        elif self.reduce_dim == 0:
            self.reduce_rate = 1
        else:
            raise ValueError('reduce dim must be greater than 0!')
        self.activation = activation
        self.which_loss = loss
        self.regularization = regularization
        self.lambda_gan = lambda_gan
        self.lambda_l1, self.lambda_edge, self.lambda_latent = lambda_recon, lambda_edge, lambda_latent
        self.batch_size = batch_size
        self.nn_input_channel = nn_input_channel
        self.dt = np.dtype(dtype)
        self.verts_ref, self.vpe = verts_ref, vpe
        self.vs = VarStore(seed)

    # ---- variables (:217-227) --------------------------------------------------------
    def _weight(self, shape):
        return self.vs.get('weights', shape, 'trunc_normal', 'conv', stddev=0.1)

    def _bias(self, shape):
        return self.vs.get('bias', shape, 'const', 'bias', value=0.1)

    def _dense(self, x, units, activation=None):
        with self.vs.scope('dense'):
            k = self.vs.get('kernel', (x.shape[-1], units), 'glorot_uniform', 'fc_kernel')
            b = self.vs.get('bias', (units,), 'zeros', 'fc_bias')
        return dense(x, k, b, activation)

    def filter(self, x, L, Fout, K):
        W = self._weight((x.shape[-1] * K, Fout))
        return chebyshev5(x, L, W, K)

    def brelu(self, x):
        if self.activation == 'b2relu':
            b = self._bias((1, x.shape[1], x.shape[2]))
        else:
            b = self._bias((1, 1, x.shape[2]))
        return bias_act(x, b, self.activation)

    # ---- composites -----------------------------------------------------------------
    def cnp(self, x, i, name):                                  # :154-171
        with self.vs.scope(name):
            x = self.filter(x, self.Laplacian[i], self.out_channels[i], self.poly_order[i])
            x = self.brelu(x)
            x = poolwT(x, self.Downsample_mtx[i])
        return x

    def udn(self, x, out_channels, i, name):                    # :173-191
        with self.vs.scope(name):
            x = poolwT(x, self.Upsample_mtx[-i - 1])
            x = self.filter(x, self.Laplacian[-i - 2], out_channels[-i - 1], self.poly_order[-i - 1])
            x = self.brelu(x)
        return x

    def cnp_d(self, x, i, name):                                # :796-810
        with self.vs.scope(name):
            x = self.filter(x, self.Laplacian_d[i], self.out_channels[i], self.poly_order_d[i])
            x = self.brelu(x)
            x = poolwT(x, self.Downsample_mtx_d[i])
        return x

    def gn(self, x, name):                                      # :681-712
        with self.vs.scope(name):
            C = x.shape[-1]
            gamma = self.vs.get('gamma', (C,), 'ones', 'gn')
            beta = self.vs.get('beta', (C,), 'zeros', 'gn')
        return group_norm(x, gamma, beta)

    def res_block(self, x_in, i, name):                         # :715-741
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

    def res_block_decoder(self, x_in, i, name):                 # :744-774
        Fi = self.out_channels[-i - 1]
        Lm = self.Laplacian[-i - 2]
        with self.vs.scope(name):
            xu = poolwT(x_in, self.Upsample_mtx[-i - 1])
            x = relu(self.gn(xu, 'group_norm'))
            with self.vs.scope('graph_linear_1'):
                x = self.filter(x, Lm, Fi // 2, 1)
            x = relu(self.gn(x, 'group_norm_1'))
            with self.vs.scope('graph_conv'):
                x = self.filter(x, Lm, Fi // 2, self.poly_order[-i - 1])
            x = relu(self.gn(x, 'group_norm_2'))
            with self.vs.scope('graph_linear_2'):
                x = self.filter(x, Lm, Fi, 1)
            if xu.shape[-1] != x.shape[-1]:
                with self.vs.scope('graph_linear_input'):
                    xu = self.filter(xu, Lm, x.shape[-1], 1)
            return x + xu

    def res_block_affine(self, x, i, name):                     # :776-793
        Lm = self.Laplacian[-i - 2]
        with self.vs.scope(name):
            x = poolwT(x, self.Upsample_mtx[-i - 1])
            with self.vs.scope('graph_conv'):
                x_gc = self.filter(x, Lm, self.out_channels[-i - 1] // 2, self.poly_order[-i - 1])
            x_gc = relu(x_gc)
            with self.vs.scope('affine'):
                x_aff = self.filter(x, Lm, x_gc.shape[-1], 1)
            return x_aff + x_gc

    # ---- networks ---------------------------------------------------------------------
    def condition(self, y, name, nz_cond, nlayers=1):           # :479-511
        y = np.asarray(y, dtype=self.dt)
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

    def cond_embeddings(self, cond, cond2):                     # :284-286
        y = self.condition(cond, 'pose', self.nz_cond, nlayers=2)
        y2 = self.condition(cond2, 'clo_label', self.nz_cond2, nlayers=self.n_layer_cond)
        return y, y2

    def encoder(self, x, y, y2):                                # :514-561
        x = np.asarray(x, dtype=self.dt)
        if self.cond_encoder:
            x = np.concatenate([x, fit_cond_dim(x, y), fit_cond_dim(x, y2)], -1)
        with self.vs.scope('generator'), self.vs.scope('encoder'):
            for i in range(len(self.out_channels)):
                if self.use_res_block:
                    x = self.res_block(x, i, 'encoder_resblock{}'.format(i + 1))
                else:
                    x = self.cnp(x, i, 'encoder_conv{}'.format(i + 1))
            if self.reduce_dim > 0:
                with self.vs.scope('1x1-conv'):
                    x = self.filter(x, self.Laplacian[-1], self.out_channels[-1] // self.reduce_rate, 1)
            x = x.reshape(x.shape[0], -1)                       # :554
            with self.vs.scope('fc_mean'):
                z_mean = self._dense(x, int(self.nz))
            with self.vs.scope('fc_var'):
                z_var = self._dense(x, int(self.nz))
        return z_mean, z_var

    def decoder_cond_vert(self, x, y, y2):                      # :564-617
        x = np.asarray(x, dtype=self.dt)
        y = np.asarray(y, dtype=self.dt)
        y2 = np.asarray(y2, dtype=self.dt)
        N = x.shape[0]
        with self.vs.scope('generator'), self.vs.scope('decoder'):
            with self.vs.scope('fc1'):
                out_nodes = int(self.p[-1] * self.out_channels[-1]) // self.reduce_rate
                x = self._dense(x, out_nodes, 'leaky_relu')
            x = x.reshape(N, int(self.p[-1]), -1)               # :584
            if self.reduce_dim > 0:
                with self.vs.scope('1x1-conv'):
                    x = self.filter(x, self.Laplacian[-1], self.out_channels[-1], 1)
            x = np.concatenate([x, fit_cond_dim(x, y), fit_cond_dim(x, y2)], -1)
            for i in range(len(self.out_channels)):
                if self.use_res_block_dec:
                    if not self.affine:
                        x = self.res_block_decoder(x, i, 'decoder_resblock_cmr{}'.format(i + 1))
                    else:
                        x = self.res_block_affine(x, i, 'decoder_resblock_affine{}'.format(i + 1))
                else:
                    x = self.udn(x, self.out_channels, i, 'decoder_conv{}'.format(i + 1))
                x = np.concatenate([x, fit_cond_dim(x, y), fit_cond_dim(x, y2)], -1)
            with self.vs.scope('outputs'):
                x = self.filter(x, self.Laplacian[0], int(self.nn_input_channel), self.poly_order[0])
                b = self._bias((1, x.shape[1], x.shape[2]))     # :615 one bias per vertex per channel
                x = x + b.astype(self.dt)
        return x

    def vae_sampling(self, z_mean, z_logvar, eps):              # :193-196 (eps injected)
        return z_mean + np.sqrt(np.exp(z_logvar)) * np.asarray(eps, dtype=self.dt)

    def generator(self, x, y, y2, eps):                         # :620-645
        z_mean, z_logvar = self.encoder(x, y, y2)
        z = self.vae_sampling(z_mean, z_logvar, eps)
        z_total = np.concatenate([z, y, y2], axis=1)
        x_hat = self.decoder_cond_vert(z_total, y, y2)
        return x_hat, z_mean, z_logvar

    def discriminator(self, x, y, y2):                          # :648-678
        x = np.asarray(x, dtype=self.dt)
        x = np.concatenate([x, fit_cond_dim(x, y), fit_cond_dim(x, y2)], -1)
        with self.vs.scope('discriminator'):
            with self.vs.scope('shared'):
                for i in range(len(self.Downsample_mtx_d)):
                    x = self.cnp_d(x, i, 'conv{}'.format(i + 1))
            with self.vs.scope('prediction_map'):
                # NB poly_order[-1], not poly_order_d (reference quirk C3, :676)
                pred_map = self.filter(x, self.Laplacian_d[-1], 1, self.poly_order[-1])
        return pred_map

    # ---- losses (:354-397) -----------------------------------------------------------
    def losses(self, g_out, g_gt, z_mean, z_logvar, d_real=None, d_fake=None, smooth=0.1):
        g_gt = np.asarray(g_gt, dtype=self.dt)
        out = {}
        diff = g_out - g_gt
        if self.which_loss == 'l1':
            out['recon'] = np.abs(diff).mean()
        elif self.which_loss == 'huber':
            a = np.abs(diff)
            out['recon'] = np.where(a <= 0.1, 0.5 * a * a, 0.1 * a - 0.5 * 0.01).mean()
        else:
            out['recon'] = (diff * diff).mean()
        lat = -0.5 * (1 + z_logvar - z_mean ** 2 - np.exp(z_logvar)).sum(axis=1)
        out['latent'] = lat.mean()
        vr = np.asarray(self.verts_ref, dtype=self.dt)
        out['edge'] = edge_loss_calc(g_out + vr, g_gt + vr, self.vpe)         # :375
        # l2_regularizer(scale) = scale*sum(w^2)/2 over dense kernels, multiplied by
        # `regularization` again (:40, :378-379; quirk C6)
        reg_g = sum(0.5 * (v.astype(self.dt) ** 2).sum() for n, v in self.vs.vars.items()
                    if self.vs.kinds[n] == 'fc_kernel' and n.startswith('generator'))
        out['fc_reg_g'] = self.regularization * self.regularization * reg_g
        out['fc_reg_d'] = self.dt.type(0.0)
        total = out['recon'] * self.lambda_l1 + out['edge'] * self.lambda_edge + \
            out['latent'] * self.lambda_latent + out['fc_reg_g']
        if d_fake is not None:
            g_labels = np.ones_like(d_fake) * (1 - smooth)
            out['gan_g'] = sigmoid_xent(d_fake, g_labels).mean()
            total = total + out['gan_g'] * self.lambda_gan
            if d_real is not None:
                d_lr = sigmoid_xent(d_real, np.ones_like(d_real) * (1 - smooth)).mean()
                d_lf = sigmoid_xent(d_fake, np.zeros_like(d_fake) + smooth).mean()
                out['gan_d'] = d_lr + d_lf
                out['loss_d'] = out['gan_d'] * self.lambda_gan + out['fc_reg_d']
        out['loss_g'] = total
        return out

"""The reference's shipped model configurations as ``CAPE(**params)`` keyword dicts, assembled
the way its entry scripts do (reference main.py:50-87, run_simple_demo.py:17-45,
config_parser.py:1-67 defaults, configs/*.yaml overrides).  The YAML files themselves are
consumed unchanged by the reference's own config_parser when this package is used as a
drop-in; this module exists for bench.py / tests where configargparse is not installed.
"""

_DEFAULTS = dict(                      # config_parser.py defaults
    num_conv_layers=8, ds_factor=2, K=2, Kd=3, nf=64, nz=18, nz_cond=24, nz_cond2=8, n_layer_cond=1,
    activation='b1leakyrelu', use_res_block=0, use_res_block_dec=1, cond_encoder=0, reduce_dim=64, affine=0,
    optim_condnet=1, batch_size=16, num_epochs=60, lr=8e-3, lr_scaler=1e-1, decay_every=1, lr_warmup=0, seed=123,
    restart=1, optimizer='sgd', loss='l1', loss_mask='', regularization=2e-3, lambda_recon=1.0, lambda_edge=1.0,
    lambda_latent=8e-4, lambda_gan=0.1,
)

_YAML = {                              # configs/<name>.yaml (fields that differ from the defaults)
    'CAPE-affineconv_nz64_pose32_clotype32_male': dict(nz=64, nz_cond=32, nz_cond2=32, affine=1, lr_warmup=1, decay_every=2),
    'CAPE-affineconv_nz64_pose32_clotype32_female': dict(nz=64, nz_cond=32, nz_cond2=32, affine=1, lr_warmup=1, decay_every=2),
    'CAPE-affineconv_nz18_pose24_clotype8_male': dict(nz=18, nz_cond=24, nz_cond2=8, affine=1, lr_warmup=1, decay_every=2),
    'CAPE-affineconv_nz18_pose24_clotype8_female': dict(nz=18, nz_cond=24, nz_cond2=8, affine=1, lr_warmup=1, decay_every=2),
    'CAPE_nz18_pose24_clotype8_male': dict(nz=18, nz_cond=24, nz_cond2=8, affine=0, lr_warmup=1, decay_every=2),
    'CAPE_nz18_pose24_clotype8_female': dict(nz=18, nz_cond=24, nz_cond2=8, affine=0, lr_warmup=1, decay_every=2),
}


def cape_params(config='CAPE-affineconv_nz64_pose32_clotype32_male', p=None, batch_size=None, decay_steps=1,
                name=None, **overrides):
    """kwargs for ``CAPE(L=, D=, U=, L_d=, D_d=, **params)``."""
    a = dict(_DEFAULTS)
    a.update(_YAML[config])
    a.update(overrides)
    nf, layers = a['nf'], a['num_conv_layers']
    if layers == 4:
        F = [nf, 2 * nf, 2 * nf, nf]
    elif layers == 6:
        F = [nf, nf, 2 * nf, 2 * nf, 4 * nf, 4 * nf]
    elif layers == 8:
        F = [nf, nf, 2 * nf, 2 * nf, 4 * nf, 4 * nf, 8 * nf, 8 * nf]
    else:
        raise NotImplementedError
    params = dict(
        F=F, K=[2] * layers, Kd=a['Kd'], p=p, nn_input_channel=3, cond_dim=14 * 9, cond2_dim=4,
        nz=a['nz'], nz_cond=a['nz_cond'], nz_cond2=a['nz_cond2'], n_layer_cond=a['n_layer_cond'],
        activation=a['activation'], use_res_block=bool(a['use_res_block']),
        use_res_block_dec=bool(a['use_res_block_dec']), cond_encoder=bool(a['cond_encoder']),
        reduce_dim=a['reduce_dim'], affine=bool(a['affine']), optim_condnet=bool(a['optim_condnet']),
        batch_size=batch_size or a['batch_size'], num_epochs=a['num_epochs'], lr=a['lr'], lr_scaler=a['lr_scaler'],
        lr_warmup=bool(a['lr_warmup']), seed=a['seed'], restart=bool(a['restart']), optimizer=a['optimizer'],
        loss=a['loss'], loss_mask=a['loss_mask'], regularization=a['regularization'],
        lambda_recon=a['lambda_recon'], lambda_edge=a['lambda_edge'], lambda_latent=a['lambda_latent'],
        lambda_gan=a['lambda_gan'], decay_steps=decay_steps, name=name or config,
    )
    if 'act_dtype' in overrides:          # extension: storage type of the mesh activations ('fp32' | 'bf16')
        params['act_dtype'] = overrides['act_dtype']
    return params

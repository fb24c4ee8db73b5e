"""Model configurations of the reference's shipped YAMLs, restated as constructor kwargs the
way run_simple_demo.py:17-45 / main.py:50-87 build them.  TEST INFRASTRUCTURE ONLY."""


def cape_params(name="affine_nz64", batch_size=16):
    nf = 64
    base = dict(
        F=[nf, nf, 2 * nf, 2 * nf, 4 * nf, 4 * nf, 8 * nf, 8 * nf],   # main.py:61
        K=[2] * 8,                                                     # main.py:65
        Kd=3, nn_input_channel=3, cond_dim=126, cond2_dim=4,
        filter='chebyshev5', activation='b1leakyrelu', pool='poolwT', unpool='poolwT',
        loss='l1', lr=8e-3, lr_scaler=0.1, lambda_gan=0.1, regularization=2e-3,
        lambda_recon=1.0, lambda_edge=1.0, lambda_latent=8e-4, batch_size=batch_size, seed=123,
        use_res_block=False, use_res_block_dec=True, cond_encoder=False, reduce_dim=64,
        n_layer_cond=1, optim_condnet=True, optimizer='sgd', decay_rate=0.99, decay_steps=1,
        momentum=0.9, num_epochs=1, restart=True, name='oracle', lr_warmup=False,
    )
    if name == "affine_nz64":     # configs/CAPE-affineconv_nz64_pose32_clotype32_male.yaml
        base.update(nz=64, nz_cond=32, nz_cond2=32, affine=True, lr_warmup=True)
    elif name == "cmr_nz18":      # configs/CAPE_nz18_pose24_clotype8_male.yaml
        base.update(nz=18, nz_cond=24, nz_cond2=8, affine=False)
    elif name == "affine_nz18":   # configs/CAPE-affineconv_nz18_pose24_clotype8_male.yaml
        base.update(nz=18, nz_cond=24, nz_cond2=8, affine=True)
    else:
        raise ValueError(name)
    return base

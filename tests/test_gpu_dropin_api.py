"""API conformance on the GPU: replay the call sequences of the reference's entry scripts against
cape_amd.models.CAPE -- run_simple_demo.py:14-49 + demos.py:367-407 (demo) and main.py:50-109 +
demos.py:47-90 (train -> test) -- with synthetic BodyData-shaped inputs."""
import copy
import os
import types

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _args_dict():
    # config_parser.py defaults overridden by configs/CAPE-affineconv_nz64_pose32_clotype32_male.yaml
    return dict(config='configs/x.yaml', name='dropin_test', num_conv_layers=8, ds_factor=2, K=2, Kd=3, nf=64, nz=64,
                nz_cond=32, nz_cond2=32, n_layer_cond=1, activation='b1leakyrelu', use_res_block=0, use_res_block_dec=1,
                cond_encoder=0, reduce_dim=64, affine=1, pose_type='rot', optim_condnet=1, batch_size=4, num_epochs=1,
                lr=8e-3, lr_scaler=1e-1, decay_every=2, lr_warmup=1, seed=123, restart=1, optimizer='sgd', loss='l1',
                loss_mask='', dataset='dataset_male_4clotypes', regularization=2e-3, lambda_recon=1.0, lambda_edge=1.0,
                lambda_latent=8e-4, lambda_gan=0.1, mode='demo', gender='male', smpl_model_folder='body_models',
                demo_n_sample=3, save_obj=0, vis_demo=0)


def _params(args_dict, p, decay_steps=1):
    # main.py:50-84 / run_simple_demo.py:17-43 verbatim in structure
    args = types.SimpleNamespace(**args_dict)
    params = copy.deepcopy(args_dict)
    params['restart'] = bool(args.restart)
    params['use_res_block'], params['use_res_block_dec'] = bool(args.use_res_block), bool(args.use_res_block_dec)
    params['nn_input_channel'] = 3
    params['K'] = [2] * args.num_conv_layers
    params['Kd'] = args.Kd
    params['p'] = p
    params['n_layer_cond'] = args.n_layer_cond
    params['cond_encoder'] = bool(args.cond_encoder)
    params['reduce_dim'] = args.reduce_dim
    params['affine'] = bool(args.affine)
    params['optimizer'] = args.optimizer
    params['lr_warmup'] = bool(args.lr_warmup)
    params['optim_condnet'] = bool(args.optim_condnet)
    params['decay_steps'] = decay_steps
    params['cond_dim'] = 126
    params['cond2_dim'] = 4
    nf = args.nf
    params['F'] = [nf, nf, 2 * nf, 2 * nf, 4 * nf, 4 * nf, 8 * nf, 8 * nf]
    for key in ['demo_n_sample', 'mode', 'dataset', 'num_conv_layers', 'ds_factor', 'nf', 'config', 'pose_type',
                'decay_every', 'gender', 'save_obj', 'vis_demo', 'smpl_model_folder']:
        params.pop(key)
    return params


def test_train_then_demo_sequences(tmp_path, mesh_ops):
    from cape_amd import models
    from cape_amd.load_data import load_graph_mtx, filter_cloth_pose
    L, D, U, p, L_ds2, D_ds2, U_ds2 = load_graph_mtx(None, load_for_demo=True)          # run_simple_demo.py:14
    ad = _args_dict()
    rng = np.random.default_rng(0)
    n_train, n_val = 8, 4
    data = types.SimpleNamespace(
        vertices_train=rng.standard_normal((n_train, 6890, 3)).astype(np.float32),
        cond1_train=rng.standard_normal((n_train, 126)).astype(np.float32),
        cond2_train=np.eye(4, dtype=np.float32)[rng.integers(0, 4, n_train)],
        vertices_val=rng.standard_normal((n_val, 6890, 3)).astype(np.float32),
        cond1_val=rng.standard_normal((n_val, 126)).astype(np.float32),
        cond2_val=np.eye(4, dtype=np.float32)[rng.integers(0, 4, n_val)])

    # ---- main.py:87-92 (train) ----
    params = _params(ad, p, decay_steps=ad['decay_every'] * n_train / ad['batch_size'])
    model = models.CAPE(L=L, D=D, U=U, L_d=L_ds2, D_d=D_ds2, project_dir=str(tmp_path), **params)
    assert (model.input_num_verts, model.nn_input_channel, model.nz, model.batch_size) == (6890, 3, 64, 4)
    model.build_graph(model.input_num_verts, model.nn_input_channel, phase='train')
    loss, t_step = model.fit(data)
    assert len(loss) == 1 and np.isfinite(loss[0]) and t_step > 0
    assert os.path.exists(os.path.join(str(tmp_path), 'checkpoints', 'dropin_test'))

    # ---- main.py:95-100 -> demos.py:63-84 (test): a NEW model object restores the checkpoint ----
    model2 = models.CAPE(L=L, D=D, U=U, L_d=L_ds2, D_d=D_ds2, project_dir=str(tmp_path), **params)
    model2.build_graph(model2.input_num_verts, model2.nn_input_channel, phase='demo')
    preds, lr_, ll_, le_ = model2.predict(data.vertices_val, data.cond1_val, data.cond2_val, labels=data.vertices_val)
    assert preds.shape == (n_val, 6890, 3) and np.isfinite(preds).all() and np.isfinite([lr_, ll_, le_]).all()
    string, *_ = model2.evaluate(data.vertices_val, data.cond1_val, data.cond2_val, data.vertices_val, model2)
    assert string.startswith('recon loss:')

    # ---- run_simple_demo.py:48-49 -> demos.py:367-407 (sample_vary_clotype) ----
    clotype = np.array([[1, 0, 0, 0], [0, 1, 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
    rot = filter_cloth_pose(mesh_ops["pack"]["demo_rot"])[0]
    rot_repeated = np.repeat(rot[np.newaxis, :], len(clotype), axis=0)
    pose_emb, clotype_emb = model2.encode_only_condition(rot_repeated, clotype)
    assert pose_emb.shape == (4, 32) and clotype_emb.shape == (4, 32)
    pose_emb = pose_emb[0]
    z_samples = np.random.normal(loc=0.0, scale=1.0, size=(ad['demo_n_sample'], model2.nz))
    for i in range(len(clotype)):
        z_sample_c = np.array([np.concatenate([s.reshape(1, -1), pose_emb.reshape(1, -1), clotype_emb[i].reshape(1, -1)],
                                              axis=1) for s in z_samples]).reshape(ad['demo_n_sample'], -1)
        predictions = model2.decode(z_sample_c, cond=pose_emb.reshape(1, -1), cond2=clotype_emb[i].reshape(1, -1))
        assert predictions.shape == (3, 6890, 3) and np.isfinite(predictions).all()
    # a model without any checkpoint refuses to run inference, like the reference's Saver.restore would
    orphan = models.CAPE(L=L, D=D, U=U, L_d=L_ds2, D_d=D_ds2, project_dir=str(tmp_path), **dict(params, name='nothing_here'))
    orphan.build_graph(6890, 3, phase='demo')
    with pytest.raises(ValueError):
        orphan.encode_only_condition(rot_repeated, clotype)
    # operator plug-points are resolved by name; unknown names fail like getattr in the reference
    with pytest.raises(AttributeError):
        models.CAPE(L=L, D=D, U=U, L_d=L_ds2, D_d=D_ds2, **dict(params, filter='no_such_filter'))


def test_tf_checkpoint_export_and_restore(tmp_path, mesh_ops):
    """A TensorFlow-format checkpoint (reference :351/:924 Saver files + `checkpoint` state) written from one model
    is found by `tf.train.latest_checkpoint`-style discovery and restored by name into another, optimiser slots
    included."""
    import torch
    from cape_amd import models, tf_checkpoint
    from cape_amd.load_data import load_graph_mtx
    L, D, U, p, L_ds2, D_ds2, U_ds2 = load_graph_mtx(None, load_for_demo=True)
    params = _params(dict(_args_dict(), name='tf_src'), p)
    src = models.CAPE(L=L, D=D, U=U, L_d=L_ds2, D_d=D_ds2, project_dir=str(tmp_path), **params)
    src.build_graph(6890, 3, phase='train')
    g = torch.Generator(device='cpu').manual_seed(5)
    for grp in ('g', 'd'):
        m = src._opt_state[grp]['m']
        m.copy_(torch.randn(m.shape, generator=g).to(m.device))
    src.global_step = 42
    src._weights_loaded = True                      # freshly initialised variables are the state to export
    prefix = src.export_tf_checkpoint(os.path.join(str(tmp_path), 'checkpoints', 'tf_dst', 'model-42'))
    reader = tf_checkpoint.BundleReader(prefix)
    assert reader.shape('generator/encoder/encoder_conv1/weights') == (6, 64)
    assert reader.shape('generator/decoder/fc1/dense/kernel') == (128, 55168)
    assert reader.has_tensor('generator/decoder/fc1/dense/kernel/Momentum')
    assert int(reader.get_tensor('training/global_step')) == 42

    dst = models.CAPE(L=L, D=D, U=U, L_d=L_ds2, D_d=D_ds2, project_dir=str(tmp_path), **dict(params, name='tf_dst'))
    dst.build_graph(6890, 3, phase='train')
    assert dst.latest_checkpoint() == prefix
    dst._weights_loaded = False
    dst._get_session()
    assert dst.global_step == 42
    a, b = src.variables(), dst.variables()
    assert set(a) == set(b)
    for k in a:
        assert np.array_equal(a[k], b[k]), k
    for grp in ('g', 'd'):
        st_a, st_b = src._opt_state[grp], dst._opt_state[grp]
        for name, (off, _) in st_a['offsets'].items():
            n = src._vars[name].numel()
            assert torch.equal(st_a['m'][off:off + n], st_b['m'][off:off + n]), name
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 6890, 3)).astype(np.float32)
    c1 = rng.standard_normal((3, 126)).astype(np.float32)
    c2 = np.eye(4, dtype=np.float32)[[0, 2, 3]]
    za, zb = src.encode(x, c1, c2), dst.encode(x, c1, c2)
    for u, v in zip(za, zb):
        assert np.array_equal(u, v)


def test_fit_runs_the_graph_runner_and_matches_eager_train_steps(tmp_path, mesh_ops):
    """fit() drives the captured HIP-graph step (the public training API must be what is benchmarked);
    its trajectory equals eager train_step() calls on the same batches and the same eps draws."""
    import torch
    from cape_amd import models
    from cape_amd.load_data import load_graph_mtx
    L, D, U, p, L_ds2, D_ds2, U_ds2 = load_graph_mtx(None, load_for_demo=True)
    B, n_train = 2, 4
    rng = np.random.default_rng(5)
    data = types.SimpleNamespace(
        vertices_train=rng.standard_normal((n_train, 6890, 3)).astype(np.float32),
        cond1_train=rng.standard_normal((n_train, 126)).astype(np.float32),
        cond2_train=np.eye(4, dtype=np.float32)[rng.integers(0, 4, n_train)],
        vertices_val=rng.standard_normal((B, 6890, 3)).astype(np.float32),
        cond1_val=rng.standard_normal((B, 126)).astype(np.float32),
        cond2_val=np.eye(4, dtype=np.float32)[rng.integers(0, 4, B)])
    ad = dict(_args_dict(), batch_size=B, lr_warmup=0, name='fit_graph')
    params = _params(ad, p, decay_steps=1000)
    a = models.CAPE(L=L, D=D, U=U, L_d=L_ds2, D_d=D_ds2, project_dir=str(tmp_path), **params)
    a.build_graph(6890, 3, phase='train')
    init = {k: v.copy() for k, v in a.variables().items()}
    np.random.seed(3)
    torch.manual_seed(3)
    a.fit(data)                                          # 2 steps of the captured adversarial step
    assert a.global_step == 4

    b = models.CAPE(L=L, D=D, U=U, L_d=L_ds2, D_d=D_ds2, project_dir=str(tmp_path), **dict(params, name='fit_eager'))
    b.build_graph(6890, 3, phase='train')
    b.load_variables(init)
    np.random.seed(3)
    torch.manual_seed(3)
    idx_g, idx_d = list(np.random.permutation(n_train)), list(np.random.permutation(n_train))
    dev = b.device
    t = lambda arr: torch.as_tensor(np.ascontiguousarray(arr), dtype=torch.float32).to(dev)
    for s_ in range(n_train // B):
        ig, id_ = idx_g[s_ * B:(s_ + 1) * B], idx_d[s_ * B:(s_ + 1) * B]
        eps = torch.zeros((B, int(b.nz)), device=dev).normal_()
        b.train_step(t(data.vertices_train[ig]), t(data.cond1_train[ig]), t(data.cond2_train[ig]), t(data.vertices_train[ig]),
                     t(data.vertices_train[id_]), t(data.cond1_train[id_]), t(data.cond2_train[id_]), eps=eps)
    for grp in ('g', 'd'):
        fa, fb = a._opt_state[grp]['flat'], b._opt_state[grp]['flat']
        d = (fa - fb).abs().max().item()
        assert d <= 1e-6 * max(fb.abs().max().item(), 1.0), (grp, d)
    # and the weights did move (the comparison is not vacuous)
    assert any(not np.array_equal(init[k], v) for k, v in b.variables().items())


def test_reference_demo_trace_on_device(tmp_path, mesh_ops):
    """The device half of tests/test_reference_entry_script.py: the call trace that the reference's UNMODIFIED
    run_simple_demo.py produced against cape_amd (constructor keywords, build_graph / encode_only_condition / decode calls and
    every array the script handed over; committed as tests/golden/run_simple_demo_trace.npz) replayed on the real model, followed
    by the demo's own post-processing (demos.py:397-407) down to the .obj files."""
    import json
    from cape_amd import models
    from cape_amd.load_data import load_graph_mtx
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "run_simple_demo_trace.npz"))
    meta = json.loads(str(g["meta"]))
    L, D, U, p, L_ds2, D_ds2, U_ds2 = load_graph_mtx(None, load_for_demo=True)
    ops_ = dict(L=L, D=D, U=U, L_d=L_ds2, D_d=D_ds2)
    # the operators the reference's own loader handed the constructor are the ones of our pack
    for k, mats in ops_.items():
        assert [list(m.shape) for m in mats] == meta["operator_shapes"][k] and [int(m.nnz) for m in mats] == meta["operator_nnz"][k], k
    ctor = dict(meta["ctor"], project_dir=str(tmp_path))
    assert ctor["p"] == list(p)
    # the demo restores a trained checkpoint (lib/models.py:209-215): give the experiment one (reference initialisers)
    trainer = models.CAPE(**ops_, **ctor)
    trainer.build_graph(trainer.input_num_verts, trainer.nn_input_channel, phase='train')
    trainer.save_checkpoint(0)

    model = models.CAPE(**ops_, **ctor)                                               # run_simple_demo.py:45
    outs = []
    for call in meta["calls"]:
        name, args = call[0], call[1:]
        if name == "build_graph":
            model.build_graph(args[0], args[1], phase=args[2])                        # run_simple_demo.py:47
        elif name == "encode_only_condition":
            pose_emb, clo_emb = model.encode_only_condition(g[args[0]], g[args[1]])     # demos.py:376
            assert pose_emb.shape == (4, model.nz_cond) and clo_emb.shape == (4, model.nz_cond2)
            assert np.isfinite(pose_emb).all() and np.isfinite(clo_emb).all()
            assert np.abs(pose_emb - pose_emb[0]).max() == 0                          # the script repeats ONE pose four times
        elif name == "decode":
            pred = model.decode(g[args[0]], cond=g[args[1]], cond2=g[args[2]])         # demos.py:395
            assert pred.shape == (3, 6890, 3) and pred.dtype == np.float32 and np.isfinite(pred).all()
            outs.append(pred)
        else:
            raise AssertionError(name)
    assert len(outs) == 4
    # demos.py:397-407: de-normalise, mask to the clothing vertices, add the minimal shape, export
    out_dir = tmp_path / "results" / "demo_results"
    os.makedirs(out_dir)
    verts = mesh_ops["pack"]["template_verts"]
    faces = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "template_faces.npy"))
    for i, pred in enumerate(outs):
        full = pred * g["train_std"] + g["train_mean"] + verts
        assert np.isfinite(full).all()
        for j in range(len(full)):
            with open(out_dir / ("clo%d_%04d.obj" % (i, j)), "w") as f:
                f.writelines("v %.8f %.8f %.8f\n" % tuple(v) for v in full[j])
                f.writelines("f %d %d %d\n" % tuple(t + 1) for t in faces)
    assert len(os.listdir(out_dir)) == 12
    # decoding the same z under another clothing type gives another mesh; the same call twice gives the same one
    assert np.abs(outs[0] - outs[1]).max() > 0
    again = model.decode(g["dec0_z"], cond=g["dec0_cond"], cond2=g["dec0_cond2"])
    assert np.array_equal(again, outs[0])


def test_reference_main_train_trace_on_device(tmp_path, mesh_ops):
    """The device half of tests/test_reference_entry_script.py::test_main_train_unmodified: the call trace the reference's
    UNMODIFIED ``main.py --mode train`` produced against cape_amd (committed as tests/golden/main_train_trace.npz) replayed on
    the real model -- constructor keywords, build_graph('train'), a REAL fit over the same synthetic BodyData (one epoch: the
    adversarial step through the captured HIP graph, validation, checkpoint), build_graph('demo') restoring that checkpoint,
    predict over the test split, then the two sampling demos' encode_only_condition / decode calls with the arrays the script
    passed, and demo_full.test_model's own error statistics (demos.py:68-80)."""
    import json
    import sys
    from cape_amd import models
    from cape_amd.load_data import load_graph_mtx
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, here)
    try:
        import entry_synth
    finally:
        sys.path.pop(0)
    g = np.load(os.path.join(here, "golden", "main_train_trace.npz"))
    meta = json.loads(str(g["meta"]))
    L, D, U, p, L_ds2, D_ds2, U_ds2 = load_graph_mtx(None, load_for_demo=True)
    ops_ = dict(L=L, D=D, U=U, L_d=L_ds2, D_d=D_ds2)
    for k, mats in ops_.items():
        assert [list(m.shape) for m in mats] == meta["operator_shapes"][k] and [int(m.nnz) for m in mats] == meta["operator_nnz"][k], k
    # the BodyData the script built from its dataset files, rebuilt from the same seeded generator
    bodydata = entry_synth.Wrapper()
    for k in entry_synth.Wrapper.FIELDS:
        assert entry_synth.summaries_match(entry_synth.summary(getattr(bodydata, k)), meta["summaries"]["bodydata." + k]), k
    np.random.seed(meta["ctor"]["seed"])                                              # main.py:11
    model = models.CAPE(**ops_, **dict(meta["ctor"], project_dir=str(tmp_path)))       # main.py:87
    first = {}
    outs, embs = [], []
    for call in meta["calls"]:
        name, args = call[0], call[1:]
        if name == "build_graph":
            model.build_graph(args[0], args[1], phase=args[2])                        # main.py:91,95
        elif name == "fit":
            first = {k: v.copy() for k, v in model.variables().items()}
            losses, t_step = model.fit(bodydata)                                      # main.py:92
            assert len(losses) == meta["ctor"]["num_epochs"] and np.isfinite(losses).all() and t_step > 0
            assert model.global_step == 2 * (entry_synth.N_TRAIN // 16) * (2 if model.bug_compat else 1)   # G and D updates per step
            moved = [k for k, v in model.variables().items() if not np.array_equal(first[k], v)]
            assert len(moved) > 0.9 * len(first)
            assert os.listdir(os.path.join(str(tmp_path), "checkpoints", meta["ctor"]["name"]))
        elif name == "predict":
            for tag, arr in zip(args[:3], (bodydata.vertices_test, bodydata.cond1_test, bodydata.cond2_test)):
                assert entry_synth.summaries_match(entry_synth.summary(arr), meta["summaries"][tag]), tag
            pred, recon, latent, edge = model.predict(data=bodydata.vertices_test, cond=bodydata.cond1_test,
                                                      cond2=bodydata.cond2_test, labels=bodydata.vertices_test, phase=args[3])
            assert pred.shape == bodydata.vertices_test.shape and np.isfinite(pred).all()
            assert all(np.isfinite(v) and v >= 0 for v in (recon, latent, edge))
            # demos.py:68-80: de-normalise and measure the per-vertex error; an untrained net on unit-variance data is O(1 sigma)
            diff = (pred - bodydata.vertices_test) * bodydata.std
            err = np.sqrt((diff ** 2).sum(axis=2))
            assert np.isfinite(err).all() and err.mean() < 10 * np.linalg.norm(bodydata.std, axis=1).mean()
        elif name == "encode_only_condition":
            pose_emb, clo_emb = model.encode_only_condition(g[args[0]], g[args[1]])     # demos.py:136,186
            assert pose_emb.shape == (len(g[args[0]]), model.nz_cond) and clo_emb.shape == (len(g[args[1]]), model.nz_cond2)
            assert np.isfinite(pose_emb).all() and np.isfinite(clo_emb).all()
            embs.append((pose_emb, clo_emb))
        elif name == "decode":
            pred = model.decode(g[args[0]], cond=g[args[1]], cond2=g[args[2]])         # demos.py:152,205
            assert pred.shape == (2, 6890, 3) and pred.dtype == np.float32 and np.isfinite(pred).all()
            outs.append(pred)
        else:
            raise AssertionError(name)
    assert len(embs) == 2 and len(outs) == meta["n_pose"] + 4
    # sample_vary_pose encodes n_pose different poses under one clothing type; sample_vary_clotype one pose under four types
    assert np.abs(embs[0][0] - embs[0][0][0]).max() > 0 and np.abs(embs[0][1] - embs[0][1][0]).max() == 0
    assert np.abs(embs[1][0] - embs[1][0][0]).max() == 0 and np.abs(embs[1][1] - embs[1][1][0]).max() > 0
    # the demo graph restored what fit saved (lib/models.py:209-215): its weights are the trained ones, not fresh initialisers
    trained = model.variables()
    assert any(not np.array_equal(first[k], v) for k, v in trained.items())

"""Pins the CPU oracle (oracle/timet_oracle.py) to vectors produced by the reference itself
(oracle/gen_golden.py -> tests/golden/*.npz).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import rel_err
from oracle import timet_oracle as O
from timetuning_amd import synth


def test_cosine_scheduler_kat(golden):
    g = golden("schedules")
    np.testing.assert_allclose(O.cosine_scheduler(0.04, 0.4, 1, 4), g["wd_1_4"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(O.cosine_scheduler(0.995, 1.0, 2, 5), g["ema_2_5"], rtol=0, atol=1e-15)
    np.testing.assert_allclose(O.cosine_scheduler(0.04, 0.4, 3, 7), g["wd_3_7"], rtol=0, atol=1e-15)
    # SURVEY 8(a) A15 known answer
    np.testing.assert_allclose(g["wd_1_4"], [0.04, 0.09272078, 0.22, 0.34727922], atol=1e-8)


def test_sinkhorn_kat(golden):
    g = golden("sinkhorn")
    kat = torch.from_numpy(g["kat_in"])
    np.testing.assert_allclose(O.sinkhorn(kat, 3).numpy(), g["kat_it3"], rtol=1e-6)
    np.testing.assert_allclose(O.sinkhorn(kat, 0).numpy(), g["kat_it0"], rtol=1e-6)
    # SURVEY 8(c) known answers
    np.testing.assert_allclose(g["kat_it3"][0], [0.40406331, 0.59593672], rtol=1e-6)
    np.testing.assert_allclose(g["kat_it0"][0], [0.2, 0.8], rtol=1e-6)


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_sinkhorn_random(golden, tag):
    g = golden("sinkhorn")
    scores = torch.from_numpy(g[f"{tag}_scores"])
    q = O.sinkhorn(torch.exp(scores / 0.05).t(), int(g[f"{tag}_iters"]))
    assert rel_err(q.numpy(), g[f"{tag}_q"]) < 1e-6
    np.testing.assert_allclose(q.sum(1).numpy(), 1.0, rtol=1e-5)


def test_sinkhorn_world2_equals_concat(golden):
    """my_utils.py:250-272 with W=2 equals the single-process solve over concatenated columns."""
    g = golden("sinkhorn_w2")
    scores = torch.from_numpy(g["scores"])
    q = O.sinkhorn(torch.exp(scores / 0.05).t(), int(g["iters"]))
    assert rel_err(q.numpy(), g["q"]) < 2e-6


def test_sinkhorn_world8_c3_shape(golden):
    """SURVEY 8(c)'s third Sinkhorn case (K, B_loc, iters, W) = (200, 8320, 10, 8) - BASELINE C3's global problem - from an 8-rank
    gloo run of the reference's my_utils.sinkhorn (my_utils.py:250-272): the single solve over the concatenated columns equals it."""
    from timetuning_amd import synth

    g = golden("sinkhorn_w8")
    scores = synth.make_sinkhorn_w8_scores()
    assert float(scores.astype(np.float64).sum()) == float(g["scores_checksum"])          # the regenerated input is the generator's, bit for bit
    W, Bl, stride = int(g["world_size"]), int(g["rows_per_rank"]), int(g["stride"])
    q = O.sinkhorn(torch.exp(torch.from_numpy(scores) / 0.05).t(), int(g["iters"])).view(W, Bl, -1)
    assert rel_err(q[:, ::stride].numpy(), g["q"]) < 5e-6
    np.testing.assert_allclose(q.double().sum((0, 1)).numpy(), g["colsum"], rtol=1e-5)
    assert abs(float(g["rowsum_min"]) - 1) < 1e-5 and abs(float(g["rowsum_max"]) - 1) < 1e-5


def test_window_mask_counts(golden):
    g = golden("label_prop")
    assert int(O.restrict_neighborhood(14, 14, 6).sum()) == int(g["mask_nnz_g14_r6"]) == 19600
    assert int(O.restrict_neighborhood(28, 28, 6).sum()) == int(g["mask_nnz_g28_r6"]) == 103684


@pytest.mark.parametrize("tag", ["a", "b", "c", "d", "e"])
def test_label_propagation(golden, tag):
    g = golden("label_prop")
    gg, fs, D, K, nlast, r, topk = [int(v) for v in g[f"{tag}_cfg"]]
    feats = torch.from_numpy(g[f"{tag}_feats"])
    q0 = torch.from_numpy(g[f"{tag}_q0"])
    seed = q0.view(gg, gg, K).permute(2, 0, 1).unsqueeze(0)
    maps = torch.stack(O.propagate_labels(nlast, r, topk, gg, feats, seed)).numpy()
    assert maps.dtype == np.float64
    assert rel_err(maps, g[f"{tag}_maps"]) < 1e-12
    assert (maps.argmax(1) == g[f"{tag}_maps"].argmax(1)).all()


def _build_from_fixture(g, teacher=False, queue=0):
    D, depth, heads, patch = [int(v) for v in g["vit_cfg"]]
    cfg = dict(embed_dim=D, depth=depth, num_heads=heads, patch_size=patch)
    K = int(g["cfg"][2])
    arch = str(g["arch"]) if "arch" in g.files else "dino-s16"
    m = O.build_oracle(arch, K, tuple(int(v) for v in g["head_list"]), mode=str(g["mode"]), vit_cfg=cfg)
    if teacher:
        m.init_momentum_teacher()
    if queue:
        m.init_queue(queue)
    return m


def test_extractor_tiny(golden):
    g = golden("timet_tiny")
    m = _build_from_fixture(g)
    bs, fs = int(g["cfg"][0]), int(g["cfg"][1])
    x = torch.from_numpy(synth.make_clips(bs, fs, 224, seed=1)).view(bs * fs, 3, 224, 224)
    with torch.no_grad():
        f, attn = m.feature_extractor(x)
        bf, _ = m.feature_extractor(x, use_head=False)
    assert rel_err(f.numpy(), g["features"]) < 1e-5
    assert rel_err(bf.numpy(), g["backbone_features"]) < 1e-5
    assert rel_err(attn[:, :, 0, :].numpy(), g["attn_cls_row"]) < 1e-5


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_extractor_other_input_sizes(golden, tag):
    """interpolate_pos_encoding's bicubic branch (dino_vision_transformer.py:219-234): 160x192 (non-square grid 10x12),
    256x256 (16x16) and 96x64 (6x4) inputs."""
    g = golden("extractor_sizes")
    D, depth, heads, patch = [int(v) for v in g["vit_cfg"]]
    cfg = dict(embed_dim=D, depth=depth, num_heads=heads, patch_size=patch)
    m = O.build_oracle("dino-s16", 20, tuple(int(v) for v in g["head_list"]), mode="stress", vit_cfg=cfg)
    H, W = [int(v) for v in g[f"{tag}_hw"]]
    x = torch.from_numpy(synth.normal(f"sizes.x.{tag}", (2, 3, H, W)))
    pos = O.interpolate_pos_encoding(m.feature_extractor.backbone["pos_embed"], (H // patch) * (W // patch), H, W, patch)
    assert rel_err(pos[0].numpy(), g[f"{tag}_pos"]) < 1e-6
    with torch.no_grad():
        f, attn = m.feature_extractor(x)
        bf, _ = m.feature_extractor(x, use_head=False)
    assert rel_err(f.numpy(), g[f"{tag}_features"]) < 1e-5
    assert rel_err(bf.numpy(), g[f"{tag}_backbone_features"]) < 1e-5
    assert rel_err(attn[:, :, 0, :].numpy(), g[f"{tag}_attn_cls_row"]) < 1e-5


def test_get_loss_internals_tiny(golden):
    g = golden("aux_tiny")
    t = golden("timet_tiny")
    m = _build_from_fixture(t)
    bs, fs, K = [int(v) for v in g["cfg"]]
    x = torch.from_numpy(synth.make_clips(bs, fs, 224, seed=1))
    loss, aux = m.get_loss(x, return_aux=True)
    assert rel_err(aux["batch_q"].numpy(), g["q"]) < 1e-4
    assert rel_err(aux["target_scores"].detach().numpy(), g["target_scores"]) < 1e-5
    assert rel_err(aux["p_map"].numpy(), g["p_map"]) < 1e-4
    mism = (aux["labels"].numpy() != g["labels"])
    # a label may flip only where the reference's own top-2 margin is at rounding level
    assert (g["label_margin"].reshape(mism.shape)[mism] < 1e-6).all()
    assert abs(loss.item() - float(t["loss0"])) < 1e-5


def _run_steps(g, teacher, queue):
    bs, fs, K, _, _, steps, E, I = [int(v) for v in g["cfg"]]
    m = _build_from_fixture(g, teacher, queue)
    opt = O.SwavOptimizerOracle(m, 1e-5, 1e-4, O.cosine_scheduler(0.04, 0.4, E, I), I, E)
    if teacher:
        m.set_momentum_teacher_schedular_params(0.995, 1.0, E, I)
    params = dict(m.named_parameters())
    assert [len(gr["params"]) for gr in opt.groups] == list(g["group_sizes"])
    use_mask = bool(int(g["use_mask"])) if "use_mask" in g.files else False
    for s in range(steps):
        if use_mask:
            x = torch.from_numpy(synth.make_smooth_clips(bs, fs, 224, seed=int(g["clip_seed0"]) + 1 + s))
        else:
            x = torch.from_numpy(synth.make_clips(bs, fs, 224, seed=1 + s))
        if use_mask and s == 0:
            with torch.no_grad():
                _, attn = m.feature_extractor(x.view(bs * fs, 3, 224, 224))
            mask0 = O.process_attentions(attn, m.feature_extractor.spatial_resolution).numpy()
            assert (mask0 == g["student_mask0"]).all() and 0 < mask0.mean() < 1
        loss = m.get_loss(x, queue_perm=g[f"perm{s}"], mask_features=use_mask)
        assert abs(loss.item() - float(g[f"loss{s}"])) < 2e-5, (s, loss.item(), float(g[f"loss{s}"]))
        opt.zero_grad()
        loss.backward()
        names = [str(n) for n in g[f"gradnorm_names{s}"]]
        mine = np.array([params[n].grad.double().norm().item() for n in names])
        np.testing.assert_allclose(mine, g[f"gradnorm{s}"], rtol=2e-4, atol=1e-9)
        for key in g.files:
            if key.startswith(f"grad{s}:"):
                assert rel_err(params[key.split(":", 1)[1]].grad.numpy(), g[key]) < 2e-4, key
        opt.step()
        m.normalize_prototypes()
        if teacher:
            m.update_momentum_teacher(opt.global_step)
        for key in g.files:
            if key.startswith(f"param{s}:"):
                assert rel_err(params[key.split(":", 1)[1]].detach().numpy(), g[key]) < 1e-6, key
        np.testing.assert_allclose([gr["lr"] for gr in opt.groups], g[f"lr{s}"], rtol=1e-12)
        np.testing.assert_allclose([gr["weight_decay"] for gr in opt.groups], g[f"wd{s}"], rtol=1e-12)
        if teacher:
            assert rel_err(m.teacher_prototypes.numpy(), g[f"teacher_prototypes{s}"]) < 1e-6
            assert rel_err(m.teacher.backbone["blocks.11.mlp.fc2.weight"].numpy(), g[f"teacher_fc2_{s}"]) < 1e-6
            assert rel_err(m.teacher.backbone["blocks.3.attn.qkv.weight"].numpy().reshape(-1)[::97], g[f"teacher_b3qkv_{s}"]) < 1e-6
        if queue:
            assert rel_err(m.queue[:64].numpy(), g[f"queue_head{s}"]) < 1e-5
            assert abs(m.queue.double().sum().item() - float(g[f"queue_sum{s}"])) < 1e-3


def test_training_steps_tiny(golden):
    _run_steps(golden("timet_tiny"), False, 0)


def test_training_steps_tiny_300_prototypes(golden):
    """K = 300 prototypes (more than 256: BASELINE C4 has 400); the reference's own run."""
    _run_steps(golden("timet_tiny_k300"), False, 0)


def test_training_steps_tiny_six_frames(golden):
    """Six-frame clips: the label propagation runs with up to five context frames per target (frame 0 + the queue of previous
    frames, mask_propagation.py:480-487); the reference's own run."""
    _run_steps(golden("timet_tiny_f6"), False, 0)


def test_training_steps_tiny_patch8(golden):
    """Patch size 8 (BASELINE C5's shape: 28 x 28 token grid, 785 tokens) against the reference's own run on a narrow ViT:
    extractor outputs, loss, gradients and parameters of two optimizer steps."""
    g = golden("timet_tiny_s8")
    m = _build_from_fixture(g)
    assert m.feature_extractor.spatial_resolution == 28
    bs, fs = int(g["cfg"][0]), int(g["cfg"][1])
    x = torch.from_numpy(synth.make_clips(bs, fs, 224, seed=1)).view(bs * fs, 3, 224, 224)
    with torch.no_grad():
        f, attn = m.feature_extractor(x)
    assert f.shape[1] == 784 and rel_err(f.numpy(), g["features"]) < 1e-5 and rel_err(attn[:, :, 0, :].numpy(), g["attn_cls_row"]) < 1e-5
    _run_steps(g, False, 0)


def test_training_steps_tiny_teacher_queue(golden):
    _run_steps(golden("timet_tiny_tq"), True, 40)


def test_training_steps_tiny_use_mask(golden):
    """--use_mask branch of get_loss (time_tuning.py:226-227,235-236,244-246,282-283,298-299)."""
    _run_steps(golden("timet_tiny_mask"), False, 0)


def test_training_steps_tiny_use_mask_teacher_queue(golden):
    _run_steps(golden("timet_tiny_mask_tq"), True, 40)


def _attn_from_cls_rows(cls):
    F_, H, N = cls.shape
    attn = torch.zeros(F_, H, N, N)
    attn[:, :, 0, :] = torch.from_numpy(cls)
    return attn


@pytest.mark.parametrize("g_", [14, 28, 7])
def test_process_attentions(golden, g_):
    """models.process_attentions (models.py:93-131) on the frames the reference itself survives (it raises IndexError
    on any frame that HAS a <= 2-pixel component, models.py:127-130: a [1,g,g] numpy mask indexes a [g,g] tensor)."""
    d = golden("attention_mask")
    ok = d[f"ref_ok_g{g_}"].astype(bool)
    assert ok.sum() >= 4
    mask = O.process_attentions(_attn_from_cls_rows(d[f"attn_cls_g{g_}"]), g_).numpy()
    assert (mask[ok] == d[f"mask_g{g_}"][ok]).all()
    assert set(np.unique(mask)) <= {0.0, 1.0}


def test_process_attentions_small_components():
    """The intended behaviour on the frames the reference cannot process: components of 1 or 2 pixels (8-connected)
    disappear, 3-pixel components and everything else stay."""
    g_ = 7
    th = np.zeros((g_, g_), np.float32)
    th[0, 0] = 1                      # single pixel
    th[0, 3] = th[1, 4] = 1           # diagonal pair
    th[3, 0] = th[4, 0] = th[5, 1] = 1  # 3 pixels, one diagonal link -> stays
    th[4:7, 4:7] = 1                  # block
    lab = O.label_components(th[None])
    sizes = sorted((lab == k).sum() for k in range(1, lab.max() + 1))
    assert sizes == [1, 2, 3, 9]


def test_gaussian_blur_properties():
    """Restated torchvision GaussianBlur(7, 0.6): normalised symmetric kernel, reflect padding keeps constants."""
    k = O.gaussian_kernel1d(7, 0.6)
    assert abs(k.sum().item() - 1) < 1e-6 and torch.allclose(k, k.flip(0)) and k.argmax().item() == 3
    np.testing.assert_allclose(k[3].item() / k[2].item(), np.exp(0.5 / 0.36), rtol=1e-5)
    x = torch.full((2, 1, 14, 14), 0.37)
    assert torch.allclose(O.gaussian_blur(x), x, atol=1e-6)
    imp = torch.zeros(1, 1, 14, 14)
    imp[0, 0, 7, 7] = 1.0
    out = O.gaussian_blur(imp)
    assert torch.allclose(out[0, 0, 4:11, 4:11], torch.outer(k, k), atol=1e-7)


@pytest.mark.parametrize("tag", ["a", "b", "c"])
def test_davis_protocol(golden, tag):
    """Evaluation protocol (mask_propagation.py:821-830): 4 context frames, neighbourhood 12, top-5, bilinear upsampling +
    arg-max, from the first frame's integer annotation."""
    d = golden("davis_protocol")
    g_, fs, D, C, R = [int(v) for v in d[f"{tag}_cfg"]]
    feats = torch.from_numpy(d[f"{tag}_feats"])
    ann = torch.from_numpy(d[f"{tag}_annotation"].astype(np.int64))
    pred, margin, maps = O.propagate_clip_predictions(4, 12, 5, g_, feats, ann, R, return_margin=True)
    assert rel_err(maps.numpy(), d[f"{tag}_maps"]) < 1e-12
    mism = pred.numpy() != d[f"{tag}_pred"]
    assert not (mism & ~d[f"{tag}_near_tie"]).any()
    j = O.jaccard(pred[-1], torch.from_numpy(np.roll(d[f"{tag}_annotation"].astype(np.int64), (2 * (fs - 1) * R // 112, 3 * (fs - 1) * R // 112), (0, 1))), C)
    assert j > 0.5  # the propagated masks follow the drifting discs


def test_to_one_hot_and_jaccard():
    y = torch.tensor([[[0, 2], [1, 2]]])
    oh = O.to_one_hot(y)
    assert oh.shape == (3, 2, 2) and oh.sum(0).eq(1).all() and oh[2, 0, 1] == 1 and oh[1, 1, 0] == 1
    a = torch.tensor([[0, 1, 1], [2, 2, 0]])
    b = torch.tensor([[0, 1, 2], [2, 2, 2]])
    # class 1: inter 1, union 2; class 2: inter 2, union 4
    assert abs(O.jaccard(a, b, 3) - 0.5) < 1e-12
    assert abs(O.jaccard(a, b, 3, involve_bg=True) - (0.5 + 0.5 + 0.5) / 3) < 1e-12


@pytest.mark.parametrize("tag", ["a", "b", "c", "d"])
def test_matched_miou(golden, tag):
    """PredsmIoU.compute (metrics.py:246-432): Hungarian, many-to-one and precision-based matching, background in / out."""
    d = golden("evaluator")
    gt, pred = d[f"{tag}_gt"].astype(int), d[f"{tag}_pred"].astype(int)
    for involve_bg in (0, 1):
        for mode, kw in dict(hungarian={}, many=dict(many_to_one=True), many_prec=dict(many_to_one=True, precision_based=True)).items():
            key = f"{tag}_{mode}_{involve_bg}"
            score, tp, fp, fn, reordered, bg = O.miou(gt, pred, involve_bg=bool(involve_bg), **kw)
            assert abs(score - float(d[key + "_score"])) < 1e-12, key
            ks = [int(k) for k in d[key + "_classes"]]
            assert [tp[k] for k in ks] == list(d[key + "_tp"]) and [fp[k] for k in ks] == list(d[key + "_fp"]) and [fn[k] for k in ks] == list(d[key + "_fn"])
            assert (reordered == d[key + "_reordered"]).all() and abs(bg - float(d[key + "_bg"])) < 1e-12


def test_proto_clustering(golden):
    d = golden("evaluator")
    x = torch.from_numpy(synth.normal("pc.x", (3, 196, 64)))
    protos = torch.from_numpy(synth.normal("pc.p", (12, 64)))
    got = O.proto_clustering(x, protos, 14, 56).numpy()
    assert not ((got != d["pc_assign"]) & ~d["pc_near_tie"]).any()


def test_state_dict_layout(golden):
    """SURVEY section 5 checkpoint layout: key names the build must reproduce."""
    keys = [str(k) for k in golden("timet_tiny_tq")["state_dict_keys"]]
    assert keys[0] == "prototypes"
    assert "feature_extractor.backbone.blocks.11.mlp.fc2.weight" in keys
    assert "feature_extractor.head.6.bias" in keys
    assert "teacher.backbone.cls_token" in keys and "teacher_prototypes" in keys


def test_full_size_c1_single_pass(golden):
    """ViT-S/16, bs 2 x 2 frames, K=50 (BASELINE config C1).  Runs the oracle with ONE backbone pass per
    extractor call (faithful=False) and checks it still reproduces the reference, which runs 4."""
    g = golden("timet_c1")
    m = _build_from_fixture(g)
    bs, fs = int(g["cfg"][0]), int(g["cfg"][1])
    x = torch.from_numpy(synth.make_clips(bs, fs, 224, seed=1))
    loss, aux = m.get_loss(x, faithful=False, return_aux=True)
    assert abs(loss.item() - float(g["loss0"])) < 2e-5
    f = aux["features"].detach().reshape(bs * fs, 196, -1)
    bf = aux["backbone_features"].reshape(bs * fs, 196, -1)
    assert rel_err(f[:, ::49, ::16].numpy(), g["features_slice"]) < 1e-5
    assert rel_err(bf[:, ::49, ::16].numpy(), g["backbone_features_slice"]) < 1e-5
    assert abs(f.double().norm().item() / float(g["features_norm"]) - 1) < 1e-6
    loss.backward()
    params = dict(m.named_parameters())
    for key in g.files:
        if key.startswith("grad0:"):
            mine = params[key.split(":", 1)[1]].grad.numpy()
            mine = mine if mine.size < 70000 else mine.reshape(-1)[::97]
            assert rel_err(mine, g[key]) < 5e-4, key


def test_standard_scaler_against_the_reference_run_with_scikit_learn(golden):
    """my_utils.normalize_and_transform's scaler stage (my_utils.py:23-30) as the reference ran it here with the REAL scikit-learn
    StandardScaler (three partial_fit batches, a constant column): the oracle's restatement on the same regenerated input."""
    from timetuning_amd import synth

    g = golden("scaler")
    z = O.standard_scale(synth.make_scaler_features())
    assert z.shape == tuple(g["shape"])
    assert np.abs(z[g["rows"]] - g["z_rows"]).max() < 2e-5          # the reference transforms in place in fp32
    assert np.abs(z.sum(0) - g["z_colsum"]).max() < 0.5 and np.abs((z ** 2).sum(0) - g["z_colsumsq"]).max() < 2.0   # of 230 000 rows
    assert (z[:, 4] == 0).all()                                      # zero-variance column: scale 1, not a division by zero


def test_pca_and_lloyd_restatements_against_scikit_learn():
    """The evaluator's two unpinned third-party pieces (faiss.PCAMatrix, faiss.Kmeans: not installed, parity unpinned) cross-checked
    against an INDEPENDENT implementation of the same published algorithms that is installed: scikit-learn's PCA (up to the sign
    of each component) and its Lloyd k-means from the same initial centroids."""
    skd = pytest.importorskip("sklearn.decomposition")
    skc = pytest.importorskip("sklearn.cluster")
    rng = np.random.default_rng(0)
    x = (rng.standard_normal((4000, 12)) @ rng.standard_normal((12, 12)) + rng.standard_normal(12) * 3).astype(np.float32)
    t, basis = O.standard_scale_pca(x, 5)
    z = O.standard_scale(x)
    pca = skd.PCA(n_components=5, svd_solver="full").fit(z)
    sign = np.sign((pca.components_ * basis).sum(1))
    assert np.abs(pca.components_ * sign[:, None] - basis).max() < 1e-8
    assert np.abs(pca.transform(z) * sign - t).max() < 1e-8
    # Lloyd from the same start: same centroids, labels and objective
    pts = np.concatenate([rng.standard_normal((300, 4)) + c for c in rng.standard_normal((6, 4)) * 6]).astype(np.float32)
    init = rng.choice(len(pts), 6, replace=False)
    cent, lab, obj = O.kmeans_lloyd(pts, init, 25)
    km = skc.KMeans(n_clusters=6, init=pts[init].astype(np.float64), n_init=1, max_iter=25, tol=0.0, algorithm="lloyd").fit(pts.astype(np.float64))
    assert np.abs(km.cluster_centers_ - cent).max() < 1e-6 and (km.labels_ == lab).all() and abs(km.inertia_ - obj) < 1e-6 * obj


def test_gaussian_blur_against_scipy():
    """The restated torchvision GaussianBlur(7, 0.6) (parity unpinned: torchvision is not installed) against an independent
    implementation of the same filter: scipy's gaussian_filter with mirror (= torch 'reflect') borders and the radius-3 kernel."""
    ndi = pytest.importorskip("scipy.ndimage")
    x = torch.from_numpy(np.random.default_rng(2).random((3, 1, 14, 14)).astype(np.float32))
    ref = np.stack([ndi.gaussian_filter(x[i, 0].numpy().astype(np.float64), sigma=float(np.float32(0.6)), mode="mirror", truncate=5.0)
                    for i in range(3)])
    assert np.abs(O.gaussian_blur(x)[:, 0].numpy() - ref).max() < 1e-6

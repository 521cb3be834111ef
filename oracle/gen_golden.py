"""TEST INFRASTRUCTURE: mint the golden vectors in tests/golden/ by RUNNING THE REAL REFERENCE.

Run in the build container (where /root/reference is mounted):

    python -m oracle.gen_golden            # writes tests/golden/*.npz

The reference modules are imported unmodified through oracle/ref_harness.py; inputs come from
oracle/cases.py (seeded).  Each file stores the reference outputs plus a sha256 of the inputs they
were computed from.  Nothing here is used by the product.
"""
import contextlib
import os
import platform
import sys
import warnings

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from oracle import cases, ref_harness as rh  # noqa: E402

OUT_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       'tests', 'golden')


def cpu_tag():
    try:
        with open('/proc/cpuinfo') as f:
            for line in f:
                if line.startswith('model name'):
                    return line.split(':', 1)[1].strip()
    except OSError:
        pass
    return platform.processor()


def meta():
    return dict(torch_version=np.array(torch.__version__), cpu=np.array(cpu_tag()),
                num_threads=np.array(torch.get_num_threads()))


def save(name, **arrays):
    os.makedirs(OUT_DIR, exist_ok=True)
    conv = {}
    for k, v in arrays.items():
        if isinstance(v, torch.Tensor):
            v = v.detach().cpu().numpy()
        conv[k] = v
    conv.update(meta())
    path = os.path.join(OUT_DIR, name + '.npz')
    np.savez_compressed(path, **conv)
    print(f'{name}: {os.path.getsize(path) / 1024:.1f} KiB')


def cfg_kwargs(cfg):
    d = cfg.as_dict()
    return d


def gen_heads(ref):
    for name in cases.HEAD_CASES:
        logits, J, cfg = cases.head_case(name)
        with rh.config(**cfg_kwargs(cfg)), torch.inference_mode():
            heads = ref.metrabs_model.MetrabsHeads(n_points=J).eval()
            heads.conv_final = torch.nn.Identity()  # run the reference's own forward on logits
            c2d, c3d = heads(logits)
        save(f'heads_{name}', coords2d=c2d, coords3d_rel=c3d,
             input_sha256=np.array(cases.sha256_of(logits)))


def gen_headconv(ref):
    for name in cases.HEADCONV_CASES:
        feat, w, b, J, cfg = cases.headconv_case(name)
        with rh.config(**cfg_kwargs(cfg)), torch.inference_mode():
            heads = ref.metrabs_model.MetrabsHeads(n_points=J).eval()
            conv = torch.nn.Conv2d(w.shape[1], w.shape[0], 1)
            conv.weight.copy_(w[:, :, None, None])
            conv.bias.copy_(b)
            heads.conv_final = conv
            c2d, c3d = heads(feat)
            logits = conv(feat)
        # fp64 evaluation of the same conv -> the reference's own rounding floor for this case
        logits64 = torch.nn.functional.conv2d(feat.double(), w.double()[:, :, None, None], b.double())
        save(f'headconv_{name}', coords2d=c2d, coords3d_rel=c3d,
             logits_absmax=np.array(float(logits.abs().max())),
             logits_fp32_vs_fp64_maxerr=np.array(float((logits.double() - logits64).abs().max())),
             input_sha256=np.array(cases.sha256_of(feat, w, b)))


def gen_parity(ref, only=None):
    """The features -> poses3d gates: the REFERENCE's own MetrabsHeads.forward (models/metrabs.py:75-85)
    and ptu3d.reconstruct_absolute (ptu3d.py:9-33) on the seeded inputs of cases.parity_gate_inputs, at
    every BASELINE config shape in three regimes, plus an fp64 evaluation of the same formulas.  The
    inputs are not stored (up to 190 MB per case): they are regenerated from the seed; their sha256
    and a float64 checksum ride along.  16-bit features: the reference's arithmetic on them is the
    f32 conv on the rounded features and the weights rounded to the feature dtype (what autocast does
    to conv_final; products of two f16 values are exact in f32)."""
    from oracle import cpu_ref
    for name, (B, C, J, hw, P, D, dtype) in cases.PARITY_GATE_SHAPES.items():
        for regime in cases.PARITY_GATE_REGIMES:
            slug = cases.parity_gate_slug(name, regime)
            if only and only not in slug:
                continue
            feat, w, b, K = cases.parity_gate_inputs(name, regime)
            wk = cases.head_weights_as_consumed(w, dtype)
            ocfg = cpu_ref.HeadConfig(proc_side=P, depth=D)
            with rh.config(**cfg_kwargs(ocfg)), torch.inference_mode():
                heads = ref.metrabs_model.MetrabsHeads(n_points=J).eval()
                conv = torch.nn.Conv2d(C, J * (1 + D), 1)
                conv.weight.copy_(wk[:, :, None, None])
                conv.bias.copy_(b)
                heads.conv_final = conv
                c2d, c3d = heads(feat.float())
                poses = rh.plain(ref.ptu3d.reconstruct_absolute(
                    c2d, c3d, K, mix_3d_inside_fov=ocfg.mix_3d_inside_fov, weak_perspective=ocfg.weak_perspective))
                logits = conv(feat.float())
                truth = cpu_ref.crop_model_from_features_fp64(feat.float(), wk, b, K, J, ocfg)
                port = cpu_ref.crop_model_from_features(feat.float(), wk, b, K, J, ocfg)
            save(slug, poses3d=poses, poses3d_fp64=truth, logits_absmax=np.array(float(logits.abs().max())),
                 median_depth_mm=np.array(float(truth[..., 2].median())),
                 reference_vs_fp64_mpjpe_mm=np.array(cpu_ref.mpjpe(poses, truth)),
                 port_vs_reference_max_mm=np.array(float((port - poses).abs().max())),
                 features_checksum=np.array(float(feat.double().sum())),
                 input_sha256=np.array(cases.sha256_of(feat, w, b, K)))


def gen_jitter(ref, only=None, runs=3):
    """How far the REFERENCE moves against itself: its MetrabsHeads.forward + reconstruct_absolute run
    `runs` more times on the inputs of every parity-gate golden (torch.linalg.lstsq and the threaded conv
    are not run-to-run deterministic) -> tests/golden/parity_reference_jitter.json: per case the largest
    |difference| (mm) and MPJPE between any two of {the stored golden, the new runs}.  The stored goldens
    are NOT rewritten.  bench.py prints these numbers beside its parity object so that "within 1e-3 mm of
    the reference" is read against the reference's own spread."""
    import itertools
    import json
    from oracle import cpu_ref
    path = os.path.join(OUT_DIR, 'parity_reference_jitter.json')
    out = json.load(open(path)) if os.path.exists(path) else {}
    for name, (B, C, J, hw, P, D, dtype) in cases.PARITY_GATE_SHAPES.items():
        for regime in cases.PARITY_GATE_REGIMES:
            slug = cases.parity_gate_slug(name, regime)
            if only and only not in slug:
                continue
            feat, w, b, K = cases.parity_gate_inputs(name, regime)
            wk = cases.head_weights_as_consumed(w, dtype)
            ocfg = cpu_ref.HeadConfig(proc_side=P, depth=D)
            stored = torch.from_numpy(np.load(os.path.join(OUT_DIR, slug + '.npz'))['poses3d'])
            results = [stored]
            with rh.config(**cfg_kwargs(ocfg)), torch.inference_mode():
                heads = ref.metrabs_model.MetrabsHeads(n_points=J).eval()
                conv = torch.nn.Conv2d(C, J * (1 + D), 1)
                conv.weight.copy_(wk[:, :, None, None])
                conv.bias.copy_(b)
                heads.conv_final = conv
                for _ in range(runs):
                    c2d, c3d = heads(feat.float())
                    results.append(rh.plain(ref.ptu3d.reconstruct_absolute(
                        c2d, c3d, K, mix_3d_inside_fov=ocfg.mix_3d_inside_fov,
                        weak_perspective=ocfg.weak_perspective)).clone())
            pairs = list(itertools.combinations(range(len(results)), 2))
            out[slug] = dict(
                runs=runs, cpu=cpu_tag(),
                run_to_run_max_mm=max(float((results[i] - results[j]).abs().max()) for i, j in pairs),
                run_to_run_mpjpe_mm=max(cpu_ref.mpjpe(results[i], results[j]) for i, j in pairs),
                new_runs_vs_stored_max_mm=max(float((results[i] - stored).abs().max()) for i in range(1, len(results))),
                new_runs_among_themselves_max_mm=max([float((results[i] - results[j]).abs().max())
                                                      for i, j in pairs if i and j] or [0.0]))
            print(slug, out[slug], flush=True)
            with open(path, 'w') as f:
                json.dump(out, f, indent=1, sort_keys=True)


def gen_recon(ref, only=None):
    for name in cases.RECON_CASES:
        if only and name != only:
            continue
        c2d, rel, K, cfg = cases.recon_case(name)
        # weak perspective: ptu.reduce_mean_masked adds a list to a torch.Size (ptu.py:30), a TypeError
        # on torch 2.x; rh.weak_perspective_runnable hands the reference a mask whose .shape adds to
        # lists -- the reference code itself runs unmodified
        ctx = rh.weak_perspective_runnable(ref) if cfg.weak_perspective else contextlib.nullcontext()
        with rh.config(**cfg_kwargs(cfg)), torch.inference_mode(), ctx:
            out = rh.plain(ref.ptu3d.reconstruct_absolute(
                c2d, rel, K, mix_3d_inside_fov=cfg.mix_3d_inside_fov,
                weak_perspective=cfg.weak_perspective))
            # the reference point itself (pre-mix), useful for debugging the solver
            inv_k = torch.linalg.inv(K)
            norm2d = (ref.ptu3d.to_homogeneous(c2d) @ inv_k.transpose(1, 2))[..., :2]
            in_fov = ref.ptu3d.is_within_fov(c2d)
            fn = (ref.ptu3d.reconstruct_ref_weakpersp if cfg.weak_perspective
                  else ref.ptu3d.reconstruct_ref_fullpersp)
            refpoint = rh.plain(fn(norm2d, rel, in_fov))
            in_fov = rh.plain(in_fov)
        save(f'recon_{name}', poses3d=out, ref_point=refpoint, in_fov=in_fov,
             input_sha256=np.array(cases.sha256_of(c2d, rel, K)))


def gen_latent(ref, only=None):
    """Row a11: the REFERENCE's Metrabs (models/metrabs.py:12-64) built with affine weights, its backbone
    an Identity over the seeded features.  The PyTorch file calls self.latent_points_to_joints (:62) but
    never defines it; the harness supplies the TF twin's definition (metrabs_tf/models/metrabs.py:80-81 ->
    tfu3d.linear_combine_points, tfu3d.py:48-49: einsum 'bjc,jJ->bJc') as that one method -- everything
    else (point counts, slicing, reconstruct_absolute, the call order) is the reference's own code."""
    import tempfile
    from oracle import cpu_ref
    for name in cases.LATENT_CASES:
        if only and name != only:
            continue
        c = cases.latent_case(name)
        with tempfile.TemporaryDirectory() as tmp:
            path = os.path.join(tmp, 'affine.npz')
            np.savez(path, w1=c['w1'].numpy(), w2=c['w2'].numpy())
            kw = dict(cfg_kwargs(c['cfg']), affine_weights=path)
            with rh.config(**kw), torch.inference_mode():
                ji = rh._JointInfoStub([f'j{i}' for i in range(c['n_joints'])], [[0, 1]])
                model = ref.metrabs_model.Metrabs(torch.nn.Identity(), ji).eval()
                assert model.heatmap_heads.n_points == c['n_raw']
                conv = torch.nn.Conv2d(c['weight'].shape[1], c['weight'].shape[0], 1)
                conv.weight.copy_(c['weight'][:, :, None, None])
                conv.bias.copy_(c['bias'])
                model.heatmap_heads.conv_final = conv
                if not hasattr(type(model), 'latent_points_to_joints'):
                    model.latent_points_to_joints = lambda points, m=model: torch.einsum(
                        'bjc,jJ->bJc', points, m.recombination_weights)
                poses = model((c['features'], c['K']))
                truth = cpu_ref.crop_model_from_features_fp64(
                    c['features'], c['weight'], c['bias'], c['K'], c['n_raw'], c['cfg'], c['w2'])
        save(f'latent_{name}', poses3d=poses, poses3d_fp64=truth,
             reference_vs_fp64_mpjpe_mm=np.array(cpu_ref.mpjpe(poses, truth)),
             features_checksum=np.array(float(c['features'].double().sum())),
             input_sha256=np.array(cases.sha256_of(c['features'], c['weight'], c['bias'], c['K'], c['w1'], c['w2'])))


def gen_warp(ref):
    for name in cases.WARP_CASES:
        c = cases.warp_case(name)
        with torch.inference_mode():
            crops = ref.warping.warp_images_with_pyramid(
                c['images'], c['K'], c['hinv'], c['dist'], c['crop_scales'],
                (c['res'], c['res']), c['image_ids'])
        save(f'warp_{name}', crops=crops,
             input_sha256=np.array(cases.sha256_of(
                 c['images_u8'], c['K'], c['hinv'], c['dist'], c['crop_scales'], c['image_ids'])))


def gen_detpre(ref):
    """Detector pre-processing + box rescale: the reference's own PersonDetector.forward with the
    network replaced by a recording stub (oracle/ref_harness.py: torchvision.resize -> its published
    F.interpolate path, ultralytics.YOLO -> recorder)."""
    import ultralytics
    det = ref.person_detector.PersonDetector()
    for name in cases.DETPRE_CASES:
        c = cases.detpre_case(name)
        ultralytics.YOLO.fake_boxes_xyxy_conf = c['net_boxes']
        with torch.inference_mode(), warnings.catch_warnings():
            warnings.simplefilter('ignore')
            boxes = det(c['images'], 0.3, 0.7, 150)
        fed = ultralytics.YOLO.last_source
        n_box = np.array([len(b) for b in boxes])
        # the full tensor is 1-2.5 MB per case: keep every 7th row / 5th column plus a digest of all
        save(f'detpre_{name}', network_input_sample=fed[:, :, ::7, ::5].contiguous(),
             network_input_shape=np.array(fed.shape), network_input_sha256=np.array(cases.sha256_of(fed)),
             boxes=torch.cat(boxes), n_box=n_box,
             input_sha256=np.array(cases.sha256_of(c['images'], *c['net_boxes'])))


def gen_filter(ref):
    """Plausibility filter + pose NMS (row f.2).  The PyTorch reference never calls these functions
    (multiperson_model.py:158-163 is commented out) and is_pose_consistent_with_box does not run as
    written (it runs given an argument for which torch.min(dim=) yields values, rh.minmax_values),
    so the vectors hold the outputs of the reference functions on the case inputs: is_pose_plausible,
    is_pose_consistent_with_box, are_augmentation_results_consistent (torch.var: unbiased),
    compute_pose_similarity, and pose_non_max_suppression given the oracle's validity mask."""
    import simplepyutils
    from oracle import cpu_ref
    pc = ref.plausibility_check
    for name in cases.FILTER_CASES:
        c = cases.filter_case(name)
        simplepyutils.mean_bones = c['mean_bones']
        ji = ref.JointInfo([f'j{i}' for i in range(c['n_joints'])], c['edges'])
        _, masks = cpu_ref.filter_poses(c['boxes'], c['poses3d'], c['poses2d'], c['edges'], c['mean_bones'])
        out = {}
        for i, (b, p3, mask) in enumerate(zip(c['boxes'], c['poses3d'], masks)):
            if len(b) == 0:
                continue
            with torch.inference_mode():
                m3 = p3.mean(dim=-3)
                out[f'plausible_{i}'] = pc.is_pose_plausible(m3, ji)
                if p3.shape[1] > 1:
                    out[f'aug_consistent_unbiased_{i}'] = pc.are_augmentation_results_consistent(p3)
                out[f'similarity_{i}'] = pc.compute_pose_similarity(m3)
                # (the PyTorch port passes torch.min's (values, indices) pair on and raises;
                #  rh.minmax_values makes torch.min / max return the values, as the TF twin's
                #  reduce_min / reduce_max do -- the rest of the function runs as written)
                m2 = c['poses2d'][i].mean(dim=-3)
                out[f'box_consistent_{i}'] = rh.plain(
                    pc.is_pose_consistent_with_box(rh.minmax_values(m2), b))
                out[f'valid_mask_{i}'] = mask
                out[f'keep_{i}'] = pc.pose_non_max_suppression(m3, b[:, 4], mask)
        save(f'filter_{name}', n_images=np.array(len(c['boxes'])),
             input_sha256=np.array(cases.sha256_of(*c['boxes'], *c['poses3d'], *c['poses2d'], c['mean_bones'])),
             **out)


def gen_backbone(ref):
    """Checkpoint format (row f.4): parameter names / shapes of the reference's backbone stack
    `Sequential(PreprocLayer(), efficientnet_v2_<size>().features)` (scripts/demo_image.py:63-66) and
    its output on name-determined weights (oracle/cases.py:deterministic_state)."""
    for size in ('s', 'l'):
        with rh.config():
            net = getattr(ref.efficientnet, f'efficientnet_v2_{size}')()
        stack = torch.nn.Sequential(ref.efficientnet.PreprocLayer(), net.features).eval()
        sd = stack.state_dict()
        stack.load_state_dict(cases.deterministic_state(sd))
        with torch.inference_mode():
            y = stack(cases.backbone_probe_input())
        save(f'backbone_effnetv2_{size}', keys=np.array(list(sd)),
             shapes=np.array([','.join(map(str, v.shape)) for v in sd.values()]), output=y)


def gen_tta(ref):
    """TTA parameter tables computed by the reference's own expressions:
    run _estimate_poses_batched with a recording stub for _predict_in_batches."""
    out = {}
    for num_aug in range(1, 7):
        rec = {}

        class CM(torch.nn.Module):
            joint_names = np.array(cases.COCO17)
            joint_edges = np.array(cases.COCO17_EDGES)
            input_resolution = 64

        est = ref.multiperson_model.Pose3dEstimator(
            CM(), {'': dict(indices=list(range(17)), names=cases.COCO17, edges=cases.COCO17_EDGES)},
            np.eye(17, dtype=np.float32))

        def fake_predict(images, K, dist, up, boxes, ibs, should_flip, rotflip, gammas, scales, aa):
            rec.update(should_flip=should_flip, rotflipmat=rotflip, gammas=gammas, scales=scales)
            n = sum(len(b) for b in boxes)
            return torch.zeros(n, len(gammas), 17, 3) + torch.tensor([0.0, 0.0, 1000.0])

        est._predict_in_batches = fake_predict
        images = torch.zeros(1, 3, 32, 32, dtype=torch.uint8)
        boxes = [torch.tensor([[4.0, 4.0, 10.0, 20.0, 1.0]])]
        with torch.inference_mode():
            est._estimate_poses_batched(
                images, boxes, torch.tensor([[[-1.0] * 3] * 3]), torch.zeros(1, 5),
                torch.eye(4)[None], torch.tensor([0.0, -1.0, 0.0]), 55, 64, 1, num_aug, True, '',
                False)
        for k, v in rec.items():
            out[f'a{num_aug}_{k}'] = v
    save('tta_params', **out)


def build_reference_estimator(ref, case):
    joint_info = rh._JointInfoStub(cases.COCO17, cases.COCO17_EDGES)
    crop_model = ref.metrabs_model.Metrabs(case['backbone'], joint_info).eval()
    conv = torch.nn.Conv2d(cases.E2E_C, 17 * (1 + case['cfg'].depth), 1)
    with torch.no_grad():
        conv.weight.copy_(case['head_w'][:, :, None, None])
        conv.bias.copy_(case['head_b'])
    crop_model.heatmap_heads.conv_final = conv
    skel = {'': dict(indices=case['skeleton'], names=[cases.COCO17[i] for i in case['skeleton']],
                     edges=[[0, 1]])}
    jtm = case['jtm'] if case['jtm'] is not None else np.eye(17, dtype=np.float32)
    est = ref.multiperson_model.Pose3dEstimator(crop_model, skel, jtm)
    if case['jtm'] is None:
        est.joint_transform_matrix = None
    return est


def gen_e2e(ref, only=None):
    import warnings
    warnings.filterwarnings('ignore', message='.*CUDA is not available.*')
    for name in cases.E2E_CASES:
        if only and name != only:
            continue
        case = cases.e2e_case(name)
        with rh.config(**cfg_kwargs(case['cfg'])), torch.inference_mode():
            est = build_reference_estimator(ref, case)
            recorded = []
            orig_get_crops = est._get_crops

            def rec_get_crops(*a, **k):
                r = orig_get_crops(*a, **k)
                recorded.append([x.clone() for x in r])
                return r

            est._get_crops = rec_get_crops
            # ... and the backbone's output of every crop-model call: with these injected in place of
            # its own backbone, an implementation's result depends on everything BUT the sampler
            # (geometry, head, reconstruction, post-processing): tests/test_gpu_e2e.py gates that
            # glue at 1e-3 mm against the poses of this same run (golden e2efeat_*)
            feats = []
            backbone = est.crop_model.backbone
            hook = backbone.register_forward_hook(lambda m, i, o: feats.append(o.detach().clone()))
            res = est._estimate_poses_batched(
                case['images'], case['boxes'], case['K'], case['dist'], case['extr'],
                case['world_up'], 55, case['ibs'], case['aa'], case['num_aug'],
                case['average_aug'], '', False)
            hook.remove()
        if name in cases.E2E_FEATURE_CASES:
            save(f'e2efeat_{name}', poses3d=torch.cat(res['poses3d']), poses2d=torch.cat(res['poses2d']),
                 n_calls=np.array(len(feats)), **{f'features_{i}': f for i, f in enumerate(feats)},
                 input_sha256=np.array(cases.sha256_of(
                     case['images'], torch.cat(case['boxes']), case['K'], case['dist'], case['extr'],
                     case['head_w'], case['head_b'])))
        if os.environ.get('MTR_GOLDEN_E2EFEAT_ONLY') == '1':
            continue
        crops0, newk0, rot0 = recorded[0]
        flat = crops0.reshape(-1, *crops0.shape[2:])
        arrays = dict(
            poses3d=torch.cat(res['poses3d']), poses2d=torch.cat(res['poses2d']),
            counts=np.array([len(p) for p in res['poses3d']]),
            batch0_crops_head=flat[:4], batch0_new_k=newk0, batch0_rot=rot0,
            batch0_crop_mean=flat.double().mean(dim=(1, 2, 3)),
            batch0_crop_sqsum=flat.double().square().sum(dim=(1, 2, 3)),
            n_internal_batches=np.array(len(recorded)),
            input_sha256=np.array(cases.sha256_of(
                case['images'], torch.cat(case['boxes']), case['K'], case['dist'], case['extr'],
                case['head_w'], case['head_b'])))
        save(f'e2e_{name}', **arrays)


def main():
    """python oracle/gen_golden.py [group | recon:case ...]   (default: every group)"""
    torch.manual_seed(0)
    ref = rh.load()
    groups = dict(heads=gen_heads, headconv=gen_headconv, recon=gen_recon, warp=gen_warp,
                  tta=gen_tta, e2e=gen_e2e, latent=gen_latent, detpre=gen_detpre, filter=gen_filter, backbone=gen_backbone,
                  parity=gen_parity, jitter=gen_jitter)
    for name in (sys.argv[1:] or [g for g in groups if g != 'jitter']):
        if ':' in name:  # one case of a group (recon, e2e: their lstsq goldens carry run-to-run jitter)
            group, only = name.split(':', 1)
            groups[group](ref, only=only)
        else:
            groups[name](ref)


if __name__ == '__main__':
    main()

"""BASELINE configs[1] at FULL size against the CPU oracle: one 25-frame 384x512 clip, 25 Karras-Euler steps, the real 1.52 B-parameter
architecture - the exact workload bench.py times (from_random(seed=42) weights, synthetic_clip(seed=1234) frames, make_noise(seed=0)).

The oracle's answer was computed once in the build container (tests/golden/make_fullsize_golden.py, ~1 h of CPU) and is committed as
tests/golden/fullsize_25step_golden.npz: final latents, per-step latent statistics, latents after steps 1 / 13 / 24, the first UNet
evaluation, the float32 VAE-encoder output, the CLIP embeddings, decoded frames (every 4th pixel + one whole frame), the wrapper's
depth and the north-star metrics of the oracle's depth / normals against a synthetic ground truth.

This is the first ORACLE comparison of the code paths that only engage at M >= 32768 rows (fused GEGLU feed-forward with its in-kernel
LayerNorm, 192-row tiles, row-split level-0 convolutions, frame-fastest temporal walks) and of the multi-stream chunk scheduling.
Reference call: /root/reference/model/depthcrafter.py:80-97.  Bounds = the trajectory bounds of tests/test_trajectory_gpu.py.
"""
import os

import numpy as np
import pytest

from util import report

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_25step_golden.npz")
GOLD16 = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fullsize_25step_fp16emu_golden.npz")


@pytest.fixture(scope="module")
def run():
    from unigeo_amd.model.depthcrafter import DepthCrafter
    from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
    from unigeo_amd.synthetic import synthetic_clip
    g = dict(np.load(GOLD))
    T, H, W, steps = (int(x) for x in g["geometry"])
    assert (T, H, W, steps) == (25, 384, 512, 25) and tuple(g["seeds"]) == (42, 1234, 0)
    pipe = DepthCrafterPipelineHIP.from_random(seed=42, workspace_bytes=40 << 30)
    try:
        eng = pipe.engine
        sample = synthetic_clip(T, H, W, seed=1234)
        frames = DepthCrafter.prepare_input(None, sample)
        nl, na = make_noise(T, H, W, seed=0)
        K = np.stack(sample["intrinsics"], 0)
        eng.set_inputs(frames, nl, na, K)
        tr = eng.run_traced(steps, 8, with_normals=True)                      # [steps,T,4,h,w] latents after every Euler step
        fr, depth, normals = eng.get_outputs(frames=True, depth=True, normals=True)
        eng.set_concurrency(3)
        eng.run(steps, 8, with_normals=True)                                   # again, untraced, VAE chunks / CLIP on three concurrent lanes: must not change a bit
        eng.set_concurrency(1)
        fr2, depth2, _ = eng.get_outputs(frames=True, depth=True, normals=False)
        eng.set_coscheduled(True)                                              # round 5: the heuristics of a context that shares the GPU with a second clip in flight
        eng.run(steps, 8, with_normals=False)
        eng.set_coscheduled(False)
        fr3, _, _ = eng.get_outputs(frames=True, depth=False, normals=False)
        cond = eng.vae_encode((frames[:2] * 2 - 1 + 0.02 * na[:2].transpose(0, 2, 3, 1)).astype(np.float16).astype(np.float32))
        emb = eng.clip_embed(frames[:4])
    finally:
        pipe.engine.close()
    return dict(g=g, tr=tr, fr=fr, depth=depth, normals=normals, fr2=fr2, depth2=depth2, fr3=fr3, cond=cond, emb=emb, K=K, na=na)


def test_conditioning_stages(run):
    g = run["g"]
    e_clip = float(np.abs(run["emb"] - g["clip_emb"][:4]).max() / np.abs(g["clip_emb"]).max())
    e_cond = float(np.abs(run["cond"] - g["cond_latents"][:2].astype(np.float32)).max() / np.abs(g["cond_latents"].astype(np.float32)).max())
    report("fullsize.clip_emb_rel_err", e_clip); report("fullsize.cond_latents_rel_err", e_cond)
    assert e_clip < 2.5e-3 and e_cond < 3e-3, (e_clip, e_cond)     # cond: the stand-alone encode call rounds its input once more than the pipeline does


def test_25_step_trajectory_against_the_oracle(run):
    g, tr = run["g"], run["tr"].astype(np.float64)
    assert tr.shape == (25, 25, 4, 48, 64) and np.isfinite(tr).all()
    amax = np.abs(tr).max(axis=(1, 2, 3, 4)); l2 = np.sqrt((tr ** 2).sum(axis=(1, 2, 3, 4))); mean = tr.mean(axis=(1, 2, 3, 4))
    e_l2 = np.abs(l2 - g["latent_l2"]) / g["latent_l2"]
    e_mean = np.abs(mean - g["latent_mean"]) / g["latent_absmax"]
    e_amax = np.abs(amax - g["latent_absmax"]) / g["latent_absmax"]
    report("fullsize.latent_l2_rel_err_max_over_steps", e_l2.max(), per_step=[float(x) for x in e_l2])
    report("fullsize.latent_mean_err_over_absmax_max_over_steps", e_mean.max())
    report("fullsize.latent_absmax_rel_err_max_over_steps", e_amax.max())
    per = []
    for j, (i, sc) in enumerate(zip(g["latents_step_index"], g["latents_step_scale"])):
        ref = g["latents_step"][j].astype(np.float64) * sc                    # stored as fp16 of latents / absmax: 5e-4 of the scale
        per.append(float(np.abs(tr[int(i)] - ref).max() / sc))
        report(f"fullsize.latent_rel_err_after_step_{int(i) + 1}", per[-1])
    e_fin = float(np.abs(tr[-1] - g["latents_final"]).max() / np.abs(g["latents_final"]).max())
    report("fullsize.latent_rel_err_final", e_fin)
    assert e_l2.max() < 1e-3 and e_mean.max() < 1e-3 and e_amax.max() < 8e-3
    assert max(per) < 4.5e-3 and e_fin < 4e-3, (per, e_fin)


def test_frames_depth_and_metrics(run):
    from oracle.geometry import prepare_output
    from unigeo_amd.harness import depth_evaluation, normal_evaluation
    import importlib.util
    spec = importlib.util.spec_from_file_location("mk", os.path.join(os.path.dirname(GOLD), "make_fullsize_golden.py"))
    mk = importlib.util.module_from_spec(spec); spec.loader.exec_module(mk)
    g, fr, depth = run["g"], run["fr"], run["depth"]
    assert fr.shape == (25, 384, 512, 3) and fr.min() >= 0 and fr.max() <= 1
    assert np.array_equal(run["fr2"], fr) and np.array_equal(run["depth2"], depth), "traced (serial) and lane-scheduled runs differ"
    # co-scheduled heuristics (ug_set_coscheduled: one fused feed-forward launch for all rows, tiles picked without the last-round fill factor): other tiles /
    # K splits, i.e. other summation orders in a few layers - over 25 steps that moves the frames as far as any rounding change does (measured 4.0e-3 against the
    # default run); the run is held to the same bound against the ORACLE as the default one
    report("fullsize.frames_abs_diff_coscheduled_vs_default", float(np.abs(run["fr3"] - fr).max()))
    e_cos = float(np.abs(run["fr3"][:, ::4, ::4] - g["frames_sub"].astype(np.float32)).max())
    report("fullsize.frames_abs_err_subsampled_coscheduled", e_cos)
    assert e_cos < 7e-3 + 5e-4, e_cos
    e_sub = float(np.abs(fr[:, ::4, ::4] - g["frames_sub"].astype(np.float32)).max())
    e_full = float(np.abs(fr[g["frames_full_index"]] - g["frames_full"].astype(np.float32)).max())
    e_mean = float(np.abs(fr[:, ::4, ::4] - g["frames_sub"].astype(np.float32)).mean())
    report("fullsize.frames_abs_err_subsampled", e_sub); report("fullsize.frames_abs_err_one_full_frame", e_full); report("fullsize.frames_mean_abs_err", e_mean)
    chm = fr.sum(-1) / 3
    report("fullsize.frames_min_err", abs(float(chm.min()) - float(g["frames_min"]))); report("fullsize.frames_max_err", abs(float(chm.max()) - float(g["frames_max"])))
    e_depth = float((np.abs(depth[:, ::4, ::4] - g["depth_sub"]) / g["depth_sub"]).max())
    report("fullsize.depth_rel_err_subsampled", e_depth)
    assert e_sub < 7e-3 + 5e-4 and e_full < 7e-3 + 5e-4, (e_sub, e_full)      # + the fp16 storage of the fixture
    # depth = 1 / (d + 0.1) with d the clip-normalised channel mean (model/depthcrafter.py:92-97): a frame error e moves d by <= 2 e / (max - min) and
    # the depth by up to 10 x that, relative - the bound follows from the frame bound; measured 9.4e-3 in round 3
    rng_ = float(g["frames_max"]) - float(g["frames_min"])
    assert e_depth < min(3e-2, 10.0 * 2.0 * (7e-3 + 5e-4) / rng_), (e_depth, rng_)
    # north_star: Abs Rel / normal mean of the HIP pipeline's depth + normals equal to the oracle's (stored) to 3 s.f.
    T, H, W = depth.shape
    gt_d = mk.synthetic_gt(T, H, W)
    _, gt_n = prepare_output(list(gt_d), list(run["K"]))
    mask = np.ones((T, H, W), bool); mask[:, :3] = False
    md = depth_evaluation(depth, gt_d, custom_mask=mask, align_with_lstsq=True)[0]
    mn = normal_evaluation(run["normals"], gt_n.numpy(), custom_mask=mask)
    want = dict(zip([str(x) for x in g["metric_names"]], g["metrics"]))
    for k, v in (("Abs Rel", md["Abs Rel"]), ("delta < 1.25", md["delta < 1.25"]), ("normal mean", mn["normal mean"]), ("normal median", mn["normal median"])):
        report(f"fullsize.metric[{k}].hip", v); report(f"fullsize.metric[{k}].oracle", want[k])
    sf3 = lambda x: float(f"{x:.3g}")
    assert md["Abs Rel"] == pytest.approx(want["Abs Rel"], rel=5e-4) and mn["normal mean"] == pytest.approx(want["normal mean"], rel=5e-4)
    assert sf3(md["Abs Rel"]) == pytest.approx(sf3(want["Abs Rel"]), rel=2e-3) and sf3(mn["normal mean"]) == pytest.approx(sf3(want["normal mean"]), rel=2e-3)


def test_depth_metric_is_sensitive_to_a_one_percent_error(run):
    """with random weights the decoded depth is unrelated to any synthetic scene, so Abs Rel / normal mean against a synthetic ground
    truth (0.18 / 89 degrees) would survive errors 100x larger than the ones this suite bounds.  Here the "ground truth" is the ORACLE's own depth
    (fixture, every 4th pixel), so the reference metric code measures exactly the HIP-vs-oracle difference: Abs Rel ~ 1e-3, delta < 1.25 = 1; and
    a spatially varying +-1 % perturbation of the HIP depth (which the scale / shift alignment of metrics/alignment.py:150-167 cannot absorb)
    must move Abs Rel to ~1e-2 - the check has the resolution the north_star tolerance needs."""
    from unigeo_amd.harness import depth_evaluation
    g, depth = run["g"], run["depth"]
    pred = np.ascontiguousarray(depth[:, ::4, ::4]).astype(np.float32)
    gt = g["depth_sub"].astype(np.float32)
    m0 = depth_evaluation(pred, gt, align_with_lstsq=True)[0]
    yy, xx = np.mgrid[0:pred.shape[1], 0:pred.shape[2]]
    checker = (((yy // 8) + (xx // 8)) % 2 * 2 - 1).astype(np.float32)[None]
    m1 = depth_evaluation(pred * (1.0 + 0.01 * checker), gt, align_with_lstsq=True)[0]
    report("fullsize.absrel_hip_vs_oracle_depth", m0["Abs Rel"]); report("fullsize.absrel_hip_vs_oracle_depth_with_1pct_checker", m1["Abs Rel"])
    assert m0["Abs Rel"] < 3e-3 and m0["delta < 1.25"] > 0.9999, m0
    assert 7e-3 < m1["Abs Rel"] < 1.5e-2 and m1["Abs Rel"] > 3.0 * m0["Abs Rel"], (m0["Abs Rel"], m1["Abs Rel"])


def test_headline_workload_against_an_fp16_run_of_the_oracle(run):
    """north_star's tolerance is stated against an fp16 REFERENCE run, at this configuration.  tests/golden/fullsize_25step_fp16emu_golden.npz (round 5,
    `make_fullsize_golden.py --fp16-storage`, 2880 s of CPU) is the oracle's answer for the same seeds with fp16 storage where the reference's fp16 pipeline has
    it: every leaf module of the UNet / CLIP tower / VAE decoder rounds its output to fp16 (fp32 arithmetic inside), the VAE encoder stays float32
    (force_upcast), latents / scaled model input / v * c of the Euler step are fp16.  The fixture itself is 2.96e-3 (final latents), 3.9e-3 (frames), 8.4e-3
    (depth) away from the fp32 fixture - an fp16 run of this network is that far from fp32 whatever executes it.  Asserted: the HIP pipeline is no further from
    fp32 than that emulation + 2 ulps of the fp16 latent grid, and HIP against the emulation stays within sqrt(2) x (two independent roundings) + the same 2 ulps."""
    g32, g16, tr = run["g"], dict(np.load(GOLD16)), run["tr"].astype(np.float64)
    assert tuple(g16["seeds"]) == (42, 1234, 0) and int(g16["fp16_storage"]) == 1 and tuple(int(x) for x in g16["geometry"]) == (25, 384, 512, 25)
    ulp = 2.0 ** -10                                                       # one unit in the last place of the fp16 latent grid, relative to max |latent| (4.9e-4 ... 9.8e-4)
    rows = []
    for j, (i, sc) in enumerate(zip(g32["latents_step_index"], g32["latents_step_scale"])):
        r32 = g32["latents_step"][j].astype(np.float64) * sc
        r16 = g16["latents_step"][j].astype(np.float64) * g16["latents_step_scale"][j]
        rows.append((int(i) + 1, float(np.abs(tr[int(i)] - r32).max() / sc), float(np.abs(r16 - r32).max() / sc), float(np.abs(tr[int(i)] - r16).max() / sc)))
    scf = float(np.abs(g32["latents_final"]).max())
    rows.append((25, float(np.abs(tr[-1] - g32["latents_final"]).max() / scf), float(np.abs(g16["latents_final"].astype(np.float64) - g32["latents_final"]).max() / scf),
                 float(np.abs(tr[-1] - g16["latents_final"]).max() / scf)))
    print("after step   HIP-vs-fp32   fp16run-vs-fp32   HIP-vs-fp16run   (max |latent error| / max |latent|)")
    for st, h32, e32, h16 in rows:
        print(f"{st:10d}   {h32:.2e}      {e32:.2e}          {h16:.2e}")
        report(f"fullsize.fp16emu.step{st}.hip_vs_fp32", h32); report(f"fullsize.fp16emu.step{st}.fp16run_vs_fp32", e32); report(f"fullsize.fp16emu.step{st}.hip_vs_fp16run", h16)
        assert h32 < e32 + 2.0 * ulp, (st, "HIP is further from fp32 than an fp16 run of the oracle + 2 fp16 ulps of the latent scale", h32, e32)
        assert h16 < 1.4143 * e32 + 2.0 * ulp, (st, "HIP vs the fp16 run of the oracle", h16, e32)
    f32, f16, fr = g32["frames_sub"].astype(np.float32), g16["frames_sub"].astype(np.float32), run["fr"][:, ::4, ::4]
    e_h32, e_1632, e_h16 = float(np.abs(fr - f32).max()), float(np.abs(f16 - f32).max()), float(np.abs(fr - f16).max())
    report("fullsize.fp16emu.frames.hip_vs_fp32", e_h32); report("fullsize.fp16emu.frames.fp16run_vs_fp32", e_1632); report("fullsize.fp16emu.frames.hip_vs_fp16run", e_h16)
    report("fullsize.fp16emu.frames.mean_abs.hip_vs_fp16run", float(np.abs(fr - f16).mean()))
    assert e_h32 < e_1632 + 2e-3 and e_h16 < 1.4143 * e_1632 + 2e-3, (e_h32, e_1632, e_h16)     # frames in [0, 1]: + 2 fp16 ulps at 1.0 (the fixtures store fp16)
    d32, d16, dh = g32["depth_sub"], g16["depth_sub"], run["depth"][:, ::4, ::4]
    r_h32, r_1632, r_h16 = float((np.abs(dh - d32) / d32).max()), float((np.abs(d16 - d32) / d32).max()), float((np.abs(dh - d16) / d16).max())
    report("fullsize.fp16emu.depth_rel.hip_vs_fp32", r_h32); report("fullsize.fp16emu.depth_rel.fp16run_vs_fp32", r_1632); report("fullsize.fp16emu.depth_rel.hip_vs_fp16run", r_h16)
    assert r_h32 < 1.5 * r_1632 + 2e-3, (r_h32, r_1632)
    # the metrics of the fp16 run of the oracle equal the fp32 oracle's (and hence HIP's, test_frames_depth_and_metrics) to 5 / 4 significant figures
    assert g16["metrics"][0] == pytest.approx(g32["metrics"][0], rel=1e-5) and g16["metrics"][2] == pytest.approx(g32["metrics"][2], rel=1e-4)


def test_three_clips_in_flight_match_the_solo_run_and_the_golden():
    """VERDICT r5 missing 2: the throughput mode bench.py reports as value_clips_in_flight - three engine contexts (own weight replica, workspace, stream, host
    thread) running CONCURRENTLY on one GPU with the co-scheduled heuristics - was only ever covered on a single context.  Here all three run the golden clip at
    the same time (unigeo_amd.shard.run_in_flight, as bench.py and harness.evaluate(models=[...]) drive it), twice: every context's frames / depth must be
    bit-identical to a solo co-scheduled run (contexts share nothing but the chip) and inside the bounds of the oracle fixture.
    Reference contract it stands in for: /root/reference/eval.py:33-39 (one clip at a time)."""
    from unigeo_amd import weights as Wt
    from unigeo_amd.model.depthcrafter import DepthCrafter
    from unigeo_amd.pipeline import DepthCrafterPipelineHIP, make_noise
    from unigeo_amd.shard import run_in_flight
    from unigeo_amd.synthetic import synthetic_clip
    g = dict(np.load(GOLD))
    T, H, W, steps = (int(x) for x in g["geometry"])
    cfgs = (Wt.UNetCfg(), Wt.VAECfg(), Wt.CLIPCfg())
    states = (Wt.random_state(Wt.unet_manifest(cfgs[0]), 42), Wt.random_state(Wt.vae_manifest(cfgs[1]), 43), Wt.random_state(Wt.clip_manifest(cfgs[2]), 44))   # = from_random(seed=42)
    pipes = [DepthCrafterPipelineHIP.from_state(*states, cfgs=cfgs, workspace_bytes=(40 << 30) if j == 0 else (12 << 30)) for j in range(3)]
    del states
    engs = [p.engine for p in pipes]
    try:
        sample = synthetic_clip(T, H, W, seed=1234)
        frames = DepthCrafter.prepare_input(None, sample)
        nl, na = make_noise(T, H, W, seed=0)
        K = np.stack(sample["intrinsics"], 0)
        for e in engs:
            e.set_coscheduled(True)
            e.set_inputs(frames, nl, na, K)
        engs[0].run(steps, 8, with_normals=False)
        fr_solo, d_solo, _ = engs[0].get_outputs(frames=True, depth=True, normals=False)
        order = []
        for rep in range(2):
            run_in_flight(3, 3, lambda i, j: (engs[j].run(steps, 8, with_normals=False), order.append(j)))
            for j, e in enumerate(engs):
                fr, d, _ = e.get_outputs(frames=True, depth=True, normals=False)
                assert np.array_equal(fr, fr_solo) and np.array_equal(d, d_solo), f"context {j}, repetition {rep}: a clip that shared the GPU differs from the solo run"
        assert sorted(order) == [0, 0, 1, 1, 2, 2]
        e_sub = float(np.abs(fr_solo[:, ::4, ::4] - g["frames_sub"].astype(np.float32)).max())
        e_depth = float((np.abs(d_solo[:, ::4, ::4] - g["depth_sub"]) / g["depth_sub"]).max())
        report("fullsize.in_flight3.frames_abs_err_subsampled", e_sub); report("fullsize.in_flight3.depth_rel_err_subsampled", e_depth)
        rng_ = float(g["frames_max"]) - float(g["frames_min"])
        assert e_sub < 7e-3 + 5e-4 and e_depth < min(3e-2, 10.0 * 2.0 * (7e-3 + 5e-4) / rng_), (e_sub, e_depth)
    finally:
        for e in engs:
            e.close()

"""GPU parity, round 5.

  * G14 (tests/golden/trajectory.npz): 32 optimisation steps of the reference's loop body (baddiffusion.py:590-615: p_losses_diffuser -> backward ->
    clip_grad_norm_(1.0) -> torch.optim.Adam -> get_cosine_schedule_with_warmup) run by the reference's own modules on SMALL_CFGS["small"]; TrainEngine
    replays it from the raw uint8 rows in both compute modes: per-step loss, pre-clip norm, LR, the weights it ends with, a forward with them, and the
    trigger-initialised DDPM images (baddiffusion.py:497-499).  The drift of the split-bf16 mode against the exact mode is reported next to both.
Tolerance: 1e-3 relative fp32 (BASELINE.json north_star) unless a tighter one is stated."""
import json
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import backdoor_ref as BD
from oracle import unet_ref as U
from tests.golden import cases as C

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda")


def make_model(cfg, seed, dev, mode=None):
    from baddiffusion_amd.unet import unet_from_config
    m = unet_from_config(cfg, **({"compute_mode": mode} if mode else {})).to(dev)
    m.load_state_dict(U.gen_params(cfg, seed))
    return m


def _replay_trajectory(mode, gpu):
    """TrainEngine over the G14 inputs; returns (losses, norms, lrs, state_dict on the CPU, model)"""
    from baddiffusion_amd.dataset import Backdoor
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    cfg = C.SMALL_CFGS["small"]
    bd = Backdoor(root=None)
    trigger = bd.get_trigger(C.TRAJ_TRIGGER, 3, cfg.sample_size)
    target = bd.get_target(C.TRAJ_TARGET, trigger)
    m = make_model(cfg, 7, gpu, mode)
    eng = TrainEngine(m, DDPMScheduler(), lr=C.TRAJ_LR, lr_warmup_steps=C.TRAJ_WARMUP, num_training_steps=C.TRAJ_TOTAL)
    u8, flags = C.traj_pool()
    u8, flags = u8.to(gpu), flags.to(gpu)
    tr, tg = trigger.to(gpu), target.to(gpu)
    losses, norms, lrs = [], [], []
    for step in range(C.TRAJ_STEPS):
        rows = C.traj_rows(step).to(gpu)
        noise, t = C.traj_noise(step)
        lrs.append(eng.current_lr())
        loss = eng.train_step(u8[rows].contiguous(), flags[rows], tr, tg, noise.to(gpu), t.to(gpu))
        losses.append(float(loss)); norms.append(float(eng.grad_norm))
    eng.close()
    return np.array(losses), np.array(norms), np.array(lrs), {k: v.detach().cpu() for k, v in m.state_dict().items()}, m, trigger, target


def test_train_engine_replays_reference_trajectory(gpu, golden):
    from baddiffusion_amd import ops
    from baddiffusion_amd.pipelines import DDPMPipeline
    from baddiffusion_amd.schedulers import DDPMScheduler
    g = golden("trajectory")
    cfg = C.SMALL_CFGS["small"]
    names = [str(n) for n in g["names"]]
    numel = {k: int(v.numel()) for k, v in U.gen_params(cfg, 7).items()}
    per_elem_move = np.array([g["pmove_final"][i] / np.sqrt(numel[k]) for i, k in enumerate(names)])
    report, final = {}, {}
    for mode in ("f32", "bf16x3"):
        losses, norms, lrs, sd, m, trigger, target = _replay_trajectory(mode, gpu)
        assert torch.equal(trigger, torch.from_numpy(g["trigger"])) and torch.equal(target, torch.from_numpy(g["target"]))
        np.testing.assert_allclose(lrs, g["lr"], rtol=1e-9, atol=1e-15)                    # optimization.py:134-138, the LR each step used
        rl = np.abs(losses - g["loss"]) / g["loss"]
        rn = np.abs(norms - g["grad_norm"]) / g["grad_norm"]
        # every one of the 32 steps; north_star's bound is 1e-3.  Round 6 (RNE hi planes, csrc/common.h): split-bf16 measures 1.6e-4 (6.0e-4 with the
        # truncated hi planes of rounds 1 - 5), exact mode 4.2e-5 -> both held to 3e-4
        assert rl.max() <= 3e-4, (mode, rl.argmax(), rl.max())
        # the pre-clip norm is the most sensitive observable: per step the two sides agree to ~1e-5, and the difference between two trajectories
        # grows as they proceed (measured, split-bf16 mode: <= 2e-5 over the first eight steps; over all 32: 1.26e-3 with RNE hi planes, round 6 --
        # 4.5e-3 at step 26 with the truncated planes before) -- held to 1e-3 while the trajectories are still one (the warm-up steps) and to
        # 3e-3 over all 32
        assert rn[: C.TRAJ_WARMUP].max() <= 1e-3 and rn.max() <= 3e-3, (mode, rn.argmax(), rn.max())
        worst = worst_rel_move = 0.0
        for i, k in enumerate(names):
            if k.endswith("key.bias"):        # gradient mathematically zero (softmax shift invariance): Adam amplifies rounding noise to +-lr steps
                continue
            got = sd[k].flatten()[:8].numpy()
            d = float(np.abs(got - g["p8_final"][i][: got.size]).max())
            worst = max(worst, d)
            worst_rel_move = max(worst_rel_move, d / max(per_elem_move[i], 1e-12))
            assert abs(float(sd[k].double().norm()) - g["pnorm_final"][i]) <= 1e-4 * g["pnorm_final"][i] + 1e-6, (mode, k)
        # The 32 steps displace an element by ~7.7e-4 (median of pmove_final / sqrt(numel)); Adam's m / sqrt(v) is a sign-like function of small
        # gradients, so a 1e-3-relative gradient difference can move single elements by a fraction of an lr per step.  Measured (profiles/
        # r06_trajectory_drift.json): exact mode 2.4e-7 (what the CPU oracle reaches too), split-bf16 mode 7.7e-7 (2.5e-6 before round 6); held
        # to 3e-6 = 0.4 % of the typical displacement.
        assert worst <= 3e-6, (mode, worst, worst_rel_move)
        x, R0, t, eps = C.train_inputs(cfg, 2)
        xn, _ = ops.qsample(x.to(gpu), R0.to(gpu), eps.to(gpu), t.to(gpu), *DDPMScheduler().device_tables(gpu))
        with torch.no_grad():
            pred = m(xn.permute(0, 3, 1, 2), t.to(gpu), return_dict=False)[0].cpu().numpy()
        np.testing.assert_allclose(pred, g["pred_final"], rtol=1e-3, atol=1e-3 * float(np.abs(g["pred_final"]).max()))
        imgs = {}
        for clip in (True, False):
            pipe = DDPMPipeline(m, DDPMScheduler(clip_sample=clip))
            init = C.traj_sample_init() + trigger.unsqueeze(0)                              # baddiffusion.py:497-499
            r = pipe(batch_size=init.shape[0], generator=torch.Generator().manual_seed(C.PIPE_SEED), init=init, output_type=None,
                     num_inference_steps=C.TRAJ_SAMPLE_STEPS)
            want = g[f"ddpm{C.TRAJ_SAMPLE_STEPS}_trigger_init_{int(clip)}"]
            err = np.abs(r.images - want)
            imgs[clip] = (float(err.max()), float(err.mean()), float((err > 1e-3).mean()))
            # five DDPM steps over 1000 training timesteps divide by sqrt(alpha_bar_t) ~ 0.05 .. 0.3 at the first steps: differences in the
            # trained weights (<= 5e-5 above) reach the images amplified; images are in [0, 1]
            # measured: exact mode max 2.7e-4, split-bf16 mode max 1.4e-3 / mean 6.7e-6 without the clip (4.7e-3 / 2.4e-5 before round 6)
            assert err.mean() <= 1e-4 and err.max() <= 3e-3, (mode, clip, imgs[clip])
        final[mode] = (losses, sd, pred, imgs)
        report[mode] = {"max_rel_loss_err": float(rl.max()), "max_rel_clipnorm_err": float(rn.max()), "worst_weight_abs_err_first8": worst,
                        "worst_weight_err_over_typical_displacement": worst_rel_move,
                        "pred_final_max_abs_err": float(np.abs(pred - g["pred_final"]).max()),
                        "trigger_init_images_max_mean_frac_gt_1e-3": {("clip" if c else "noclip"): v for c, v in imgs.items()}}
    # drift of the split-bf16 mode against the exact mode over the 32 steps
    lf, sf, pf, _ = final["f32"]
    lb, sb, pb, _ = final["bf16x3"]
    wd = max(float((sf[k] - sb[k]).abs().max()) for k in sf if not k.endswith("key.bias"))
    report["bf16x3_vs_f32"] = {"max_rel_loss_diff": float((np.abs(lf - lb) / lf).max()), "max_weight_abs_diff": wd,
                               "pred_final_max_abs_diff": float(np.abs(pf - pb).max())}
    assert report["bf16x3_vs_f32"]["max_rel_loss_diff"] <= 4e-4 and wd <= 1.0 * C.TRAJ_LR
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    with open(os.path.join(ROOT, "gpurun_out", "r06_trajectory_drift.json"), "w") as f:
        json.dump(report, f, indent=1)
    print("G14 drift report:", json.dumps(report))


def test_static_cache_is_reset_by_in_place_weight_changes(gpu):
    """inside ONE static_weights() block the prepared weight planes are keyed on (params pointer, workspace pointer, B) only;
    load_state_dict on the same buffer, bd_unet_set_compute_mode and bd_unet_reset_static_cache must each make the next forward re-read the weights."""
    from baddiffusion_amd.unet import unet_from_config
    cfg = C.SMALL_CFGS["small"]
    m = unet_from_config(cfg).to(gpu)
    P = U.gen_params(cfg, 4)
    m.load_state_dict(P)
    x = torch.randn(4, 3, cfg.sample_size, cfg.sample_size, generator=torch.Generator().manual_seed(2)).to(gpu)
    with torch.no_grad(), m.static_weights():
        a = m(x, 11).sample.clone()
        assert torch.equal(m(x, 11).sample, a)                       # the cached planes are used, same result
        m.load_state_dict({k: v * 1.05 for k, v in P.items()})       # same flat buffer, new values
        b = m(x, 11).sample.clone()
        assert not torch.equal(a, b)
        with torch.no_grad():
            m.flat.data.mul_(1.0 / 1.05)          # an in-place change the Python side cannot see: the prepared conv / attention planes are stale
        stale = m(x, 11).sample.clone()           # (biases / GroupNorm parameters are read from the buffer directly: a mixture, by contract undefined)
        m._reset_static_cache()                   # explicit reset: everything is re-read
        c = m(x, 11).sample
        assert float((c - a).abs().max()) < 1e-4 * float(a.abs().max()) and not torch.equal(c, b)
        assert float((stale - a).abs().max()) > 10 * float((c - a).abs().max())
        m.set_compute_mode("f32") if hasattr(m, "set_compute_mode") else None


def test_tune_set_slot_count_keeps_results(gpu):
    """bd_tune_set("ps_wg3_slots", n) (bench.py --gpus N sweeps it inside one process): another K-split of the large weight gradients changes the
    fp32 summation order only, workspaces are re-laid out, and 0 restores the default."""
    import ctypes
    from baddiffusion_amd import _lib as L
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    lib = L.load()
    cfg = U.CIFAR10_32
    g = torch.Generator().manual_seed(9)
    B = 16
    u8 = torch.randint(0, 256, (B, 32, 32, 3), generator=g, dtype=torch.uint8).to(gpu)
    pois = (torch.arange(B) % 4 == 0).to(gpu)
    from baddiffusion_amd.dataset import Backdoor
    trig = Backdoor(root=None).get_trigger("BOX_14", 3, 32); tgt = Backdoor(root=None).get_target("CORNER", trig)
    eps = torch.randn(B, 3, 32, 32, generator=g).to(gpu); t = torch.randint(0, 1000, (B,), generator=g).to(gpu)
    res = {}
    try:
        for slots in (0, 96, 0):
            assert lib.bd_tune_set(b"ps_wg3_slots", slots) == 0
            m = make_model(cfg, 0, gpu)
            eng = TrainEngine(m, DDPMScheduler(), lr=2e-4)
            loss = eng.train_step(u8, pois, trig.to(gpu), tgt.to(gpu), eps, t)
            res.setdefault(slots, []).append((float(loss), eng.grads.clone(), m.workspace_bytes(B, True)))
            eng.close()
    finally:
        lib.bd_tune_set(b"ps_wg3_slots", 0)
    (l0, g0, w0), (l0b, g0b, w0b) = res[0]
    (l1, g1, w1), = res[96]
    assert l0 == l0b and torch.equal(g0, g0b) and w0 == w0b          # 0 restores the default exactly
    assert l1 == l0                                                    # forward untouched (the op workspace is a maximum over all launches: it may not move)
    assert float((g1 - g0).norm() / g0.norm()) < 1e-5 and not torch.equal(g1, g0)
    assert lib.bd_tune_set(b"no_such_knob", 1) != 0
    # round 6: ops.tune_set drops every live model's pooled workspaces (callers used to clear model._ws_pool themselves), and a knob
    # change BETWEEN a training forward and its backward is refused instead of moving the saved activations under the live workspace
    from baddiffusion_amd import ops
    m = make_model(cfg, 0, gpu)
    x = torch.randn(2, 32, 32, 3, device=gpu); tt = torch.tensor([3, 500], device=gpu)
    try:
        pred, ws = m._run_forward(m.flat.data, x, tt, training=True)
        m._release_ws(ws)
        assert any(m._ws_pool.values())
        ops.tune_set("ps_wg3_slots", 96)
        assert not any(m._ws_pool.values())
        grads = torch.empty_like(m.flat.data); lo, hi = ctypes.c_int64(), ctypes.c_int64()
        rc = lib.bd_unet_backward_segment(m._plan, 0, 2, m.flat.data.data_ptr(), x.data_ptr(), 3, pred.data_ptr(), pred.shape[-1], grads.data_ptr(),
                                          ws.data_ptr(), ws.numel(), L.stream(), ctypes.byref(lo), ctypes.byref(hi))
        assert rc != 0 and b"bd_tune_set" in lib.bd_last_error()
        with pytest.raises(RuntimeError):
            ops.tune_set("no_such_knob", 1)
    finally:
        ops.tune_set("ps_wg3_slots", 0)
    torch.cuda.synchronize()


def test_rccl_transport_is_agreed_on_collectively(gpu):
    """a rank on which the direct RCCL transport cannot be opened must not leave the others inside RCCL's bootstrap.  With the failure
    injected (BD_RCCL_FAIL_RANK, honoured only with BD_TEST_HOOKS=1) at EACH host-side stage of TrainEngine._open_rccl -- library load, unique id,
    ncclCommInitRank, self-test (an agreement behind every stage, not only the first) -- the engine reports the reason and runs the
    c10d transport; the step is bit-identical to the ordinary one.  Without BD_TEST_HOOKS the variable is ignored (product path)."""
    import subprocess, sys

    def run(fail, hooks, batch):
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", BD_DP_TRANSPORT="rccl", BD_RCCL_FAIL_RANK=fail, BD_T_BATCH=str(batch))
        env.pop("BD_TEST_HOOKS", None)
        if hooks:
            env["BD_TEST_HOOKS"] = "1"
        r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "_force_dp_worker.py")], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])

    for fail, batch in (("0", 128), ("0:uid", 8), ("0:init", 8), ("0:selftest", 8)):
        d = run(fail, True, batch)
        assert d["transport"] == "c10d:nccl" and "fault injection" in (d.get("transport_note") or "") and d["equal"] and d["moments_equal"], (fail, d)
    d = run("0", False, 8)
    assert d["transport"] == "rccl-direct" and d["transport_note"] is None and d["equal"], d

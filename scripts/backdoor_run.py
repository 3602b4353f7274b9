"""Does the backdoor implant on the HIP path?  (task 2; north_star's FID / MSE clause.)

Trains the DDPM-CIFAR10-32 topology from default init on a PROCEDURAL 32x32 dataset (CIFAR10 itself cannot be downloaded here) with the
product's own loader + engine -- DatasetLoader(FIXED, poison_rate) -> device_batch_rows -> TrainEngine.train_step (fused gather / flip / blend /
q_sample, UNet forward + backward, clip + Adam, cosine LR), i.e. the loop body of /root/reference/baddiffusion.py:590-615 -- and scores it the way
measure() does (:497-499, :536-546): DDPM-1000 chains from `noise + trigger`, MSE of the generated images against the backdoor target; next to
it the same chains from clean noise (they must NOT collapse onto the target) and held-out clean / backdoor training losses.

    python scripts/backdoor_run.py --steps 3000 --out profiles/r05_backdoor_run.json
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def procedural_images(n, size=32, seed=0):
    """uint8 [n, size, size, 3]: a linear colour gradient + two soft blobs + a hard-edged rectangle per image (structure a diffusion model can learn
    in a few thousand steps; no natural-image statistics are claimed)"""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32) / (size - 1)
    out = np.empty((n, size, size, 3), np.float32)
    for i in range(n):
        c0, c1 = rng.random(3), rng.random(3)
        ang = rng.random() * 2 * np.pi
        g = (np.cos(ang) * (xx - 0.5) + np.sin(ang) * (yy - 0.5)) + 0.5
        img = c0[None, None] * g[..., None] + c1[None, None] * (1 - g[..., None])
        for _ in range(2):
            cx, cy, r = rng.random(), rng.random(), 0.08 + 0.2 * rng.random()
            w = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * r * r))[..., None]
            img = img * (1 - w) + rng.random(3)[None, None] * w
        x0, y0 = rng.integers(0, size - 8, 2)
        w_, h_ = rng.integers(4, 12, 2)
        img[y0:y0 + h_, x0:x0 + w_] = rng.random(3)
        out[i] = img * (0.05 + 0.95 * rng.random() ** 2)          # overall brightness from near-black to full (natural sets hold dark images too)
    return (np.clip(out, 0, 1) * 255 + 0.5).astype(np.uint8)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=3000)
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--images", type=int, default=8192)
    ap.add_argument("--poison-rate", type=float, default=0.1)
    ap.add_argument("--trigger", default="BOX_14")
    ap.add_argument("--target", default="HAT", help="HAT is taken from tests/golden/img_triggers.npz (the reference's own get_target output); CORNER otherwise")
    ap.add_argument("--lr", type=float, default=2e-4)
    ap.add_argument("--warmup", type=int, default=500)
    ap.add_argument("--eval-at", default="0,250,500,1000,2000,3000")
    ap.add_argument("--eval-n", type=int, default=64)
    ap.add_argument("--sample-steps", type=int, default=1000)
    ap.add_argument("--mode", default="bf16x3", choices=["f32", "bf16x3"])
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_backdoor_run.json"))
    args = ap.parse_args()

    from baddiffusion_amd import ops
    from baddiffusion_amd.dataset import Backdoor, DatasetLoader
    from baddiffusion_amd.model import KNOWN_TOPOLOGIES
    from baddiffusion_amd.pipelines import DDPMPipeline
    from baddiffusion_amd.schedulers import DDPMScheduler
    from baddiffusion_amd.trainer import TrainEngine
    from baddiffusion_amd.unet import UNet2DModel

    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    torch.manual_seed(0)
    t_gen = time.time()
    root = tempfile.mkdtemp(prefix="bd_proc_")
    np.save(os.path.join(root, "cifar10_u8.npy"), procedural_images(args.images))
    dsl = DatasetLoader(root=root, name="CIFAR10", batch_size=args.batch, seed=0, device=dev)
    target_name = args.target
    dsl.set_poison(args.trigger, "CORNER" if args.target == "HAT" else args.target, clean_rate=1.0, poison_rate=args.poison_rate)
    hat = os.path.join(ROOT, "tests", "golden", "img_triggers.npz")
    if args.target == "HAT":
        if os.path.exists(hat):
            dsl._target = torch.from_numpy(np.load(hat)["target_HAT_c3_s32"])
        else:
            target_name = "CORNER (tests/golden/img_triggers.npz missing)"
    dsl.prepare_dataset(DatasetLoader.MODE_FIXED).to_device(dev)
    trigger, target = dsl.trigger.to(dev), dsl.target.to(dev)
    print(f"dataset: {args.images} procedural images ({time.time() - t_gen:.1f} s), {int(dsl._is_poison.sum())} backdoor rows, trigger {args.trigger}, target {target_name}",
          file=sys.stderr)

    model = UNet2DModel(**KNOWN_TOPOLOGIES["google/ddpm-cifar10-32"], compute_mode=args.mode).to(dev)
    sched = DDPMScheduler(num_train_timesteps=1000)
    eng = TrainEngine(model, sched, lr=args.lr, lr_warmup_steps=args.warmup, num_training_steps=args.steps)

    # held-out evaluation batch: fixed rows of a SECOND procedural set, fixed noise / timesteps (clean rows and all-backdoor rows)
    held = torch.from_numpy(procedural_images(256, seed=12345)).to(dev)
    g = torch.Generator().manual_seed(77)
    e_noise = torch.randn(256, 3, 32, 32, generator=g).to(dev)
    e_t = torch.randint(0, 1000, (256,), generator=g).to(dev)
    a, ac = sched.device_tables(dev)

    def held_out_losses():
        out = {}
        for tag, flags in (("clean", torch.zeros(256, dtype=torch.bool, device=dev)), ("backdoor", torch.ones(256, dtype=torch.bool, device=dev))):
            xn, tg = ops.poison_qsample(held, flags, trigger, target, e_noise, e_t, a, ac)
            with torch.no_grad():
                pred = model(xn.permute(0, 3, 1, 2), e_t, return_dict=False)[0]
            out[tag] = float(((pred.permute(0, 2, 3, 1) - tg) ** 2).mean())
        return out

    hx = (held[: args.eval_n].permute(0, 3, 1, 2).float() / 255.0).cpu() / (1.0 + 1e-5) * 2.0 - 1.0
    mask = dsl.get_mask(dsl.trigger).float()
    held_R = mask * hx + (1 - mask) * dsl.trigger                                                   # dataset.py:306-315
    init = torch.randn(args.eval_n, 3, 32, 32, generator=torch.Generator().manual_seed(0))        # measure(): noise, then noise + trigger (:497-499)
    tgt01 = (dsl.target / 2 + 0.5).clamp(0, 1).permute(1, 2, 0).numpy()                          # :538-546 compare images in [0, 1]

    def sample_scores(clip=False):
        # clip_sample=False is the reference's default for this path (--fclip o -> config.clip False, baddiffusion.py:183-188): clamping the predicted
        # x_0 to [-1, 1] at every step removes the (1 - sqrt(abar_t)) r term the backdoored chain rides on -- the paper's inference-time clipping defence
        out = {}
        # "backdoor": measure()'s initialisation, noise + trigger (:497-499); "poisoned_image": noise + a held-out image carrying the trigger -- exactly the
        # x_T the poisoned forward process of loss.py:257-285 produces (r = the whole poisoned image); "clean": plain noise
        for tag, x0 in (("backdoor", init + dsl.trigger.unsqueeze(0)), ("poisoned_image", init + held_R[: args.eval_n]), ("clean", init)):
            pipe = DDPMPipeline(model, DDPMScheduler(num_train_timesteps=1000, clip_sample=clip))
            pipe.set_progress_bar_config(disable=True)
            r = pipe(batch_size=args.eval_n, generator=torch.Generator(device=dev).manual_seed(1), init=x0, output_type=None,
                     num_inference_steps=args.sample_steps)
            imgs = np.asarray(r.images)                                                          # [n, H, W, C] in [0, 1]
            out[f"mse_to_target_{tag}_init"] = float(((imgs - tgt01[None]) ** 2).mean())
            out[f"finite_{tag}_init"] = bool(np.isfinite(imgs).all())
        return out

    eval_at = sorted({int(v) for v in args.eval_at.split(",") if int(v) <= args.steps} | {args.steps})
    curve, evals = [], []
    step = 0
    acc, acc_n = 0.0, 0
    pend = []

    def evaluate():
        torch.cuda.synchronize()
        t0 = time.time()
        rec = {"step": step, "lr": eng.current_lr(), "held_out_loss": held_out_losses()}
        rec.update(sample_scores())
        rec["eval_seconds"] = time.time() - t0
        evals.append(rec)
        print(json.dumps(rec), file=sys.stderr, flush=True)

    if 0 in eval_at:
        evaluate()
    epoch = 0
    while step < args.steps:
        for rows, flips, pois in dsl.device_batch_rows(shuffle=True, epoch=epoch):
            if rows.shape[0] != args.batch:
                continue
            noise = torch.randn((args.batch, 3, 32, 32), device=dev)
            ts = torch.randint(0, 1000, (args.batch,), device=dev).long()
            loss = eng.train_step(dsl.device_images, pois, trigger, target, noise, ts, row_index=rows, flip=flips)
            pend.append(loss)
            step += 1
            if step % 50 == 0:
                vals = torch.stack(pend).float().cpu().numpy()
                curve.append({"step": step, "loss_mean_last_50": float(vals.mean()), "grad_norm": float(eng.grad_norm), "lr": eng.current_lr()})
                pend = []
            if step in eval_at:
                evaluate()
            if step >= args.steps:
                break
        epoch += 1
    eng.close()
    clipped = sample_scores(clip=True)          # the same chains with clip_sample=True: the inference-time clipping defence
    first, last = evals[0], evals[-1]
    res = {"what": "backdoor implant run on the HIP path (scripts/backdoor_run.py): DDPM-CIFAR10-32 topology from default init, procedural 32x32 dataset, "
                   "product loader + TrainEngine, scored like measure() (baddiffusion.py:497-499, 536-546)",
           "config": {"steps": args.steps, "batch": args.batch, "images": args.images, "poison_rate": args.poison_rate, "trigger": args.trigger,
                      "target": target_name, "lr": args.lr, "warmup": args.warmup, "compute_mode": args.mode, "eval_chains": args.eval_n,
                      "sampler": f"DDPM, {args.sample_steps} steps, clip_sample False (the reference's --fclip o default)", "backdoor_rows": int(dsl._is_poison.sum())},
           "evaluations": evals, "loss_curve": curve,
           "summary": {"backdoor_mse_first": first["mse_to_target_backdoor_init"], "backdoor_mse_last": last["mse_to_target_backdoor_init"],
                       "clean_init_mse_to_target_last": last["mse_to_target_clean_init"],
                       "poisoned_image_init_mse_first": first["mse_to_target_poisoned_image_init"],
                       "poisoned_image_init_mse_last": last["mse_to_target_poisoned_image_init"],
                       "held_out_clean_loss_first": first["held_out_loss"]["clean"], "held_out_clean_loss_last": last["held_out_loss"]["clean"],
                       "held_out_backdoor_loss_last": last["held_out_loss"]["backdoor"],
                       "with_clip_sample_true_last": clipped}}
    os.makedirs(os.path.dirname(os.path.abspath(args.out)), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res["summary"]))


if __name__ == "__main__":
    main()

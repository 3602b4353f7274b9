"""Adversarial Neuron Pruning defense on the MI355X path -- the counterpart of the reference's anp_defense.py / anp_config.py / anp_util.py.

    python anp_defense.py --ckpt <backdoored checkpoint dir> [--epoch 10] [--learning_rate 1e-4] [--perturb_budget 4.0] [--lr_sched]
                          [--output_dir DIR] [--tag T] [--batch 128]

Same flags, output-directory naming (anp_config.py:48-51), loop (anp_defense.py:113-186) and score file as the reference; the compute is
baddiffusion_amd/anp.py (perturbed UNet = the HIP plan on effective weights).  Trackers (wandb / tensorboard) are out of scope;
`--ckpt` must be a local diffusers-layout directory with the training run's args.json (trigger, target, poison_rate, dataset), as upstream.
"""
import argparse
import json
import math
import os
from dataclasses import dataclass
from typing import Union

import numpy as np
import torch

from baddiffusion_amd import anp, ops
from baddiffusion_amd.dataset import Backdoor, DatasetLoader
from baddiffusion_amd.model import DiffuserModelSched, batch_sampling
from baddiffusion_amd.pipelines import DDPMPipeline


@dataclass
class Config:                                        # anp_config.py:10-43
    project: str = "anp_test"
    dataset_path: Union[str, os.PathLike] = "datasets"
    dataset: str = "CIFAR10"
    batch: int = 128
    epoch: int = 10
    trigger: str = Backdoor.TRIGGER_NONE
    target: str = Backdoor.TARGET_TG
    poison_rate: float = None
    ckpt: Union[str, os.PathLike] = None
    clip: bool = True
    learning_rate: float = 1e-4
    lr_sched: bool = False
    gpu: str = "0"
    perturb_budget: float = 4.0
    tag: str = None
    measure_sample_n: int = 128
    eval_sample_n: int = 16
    eval_max_batch: int = 256
    save_image_epochs: int = 1
    output_dir: Union[str, os.PathLike] = ""
    measure_dir: Union[str, os.PathLike] = "measure"
    score_file: Union[str, os.PathLike] = "score.json"
    lr_warmup_steps: int = 500
    seed: int = 0
    num_images: int = None                           # (synthetic / truncated dataset: tests)


def naming_fn(config):                               # anp_config.py:48-51
    add_on = "_sched" if config.lr_sched else ""
    add_on += f"_{config.tag}" if config.tag is not None else ""
    return f"res_anp_{config.epoch}_lr{config.learning_rate}_pb{config.perturb_budget}{add_on}_{os.path.basename(str(config.ckpt).rstrip('/'))}"


def get_config(argv=None):                           # anp_config.py:53-99
    config = Config()
    p = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    p.add_argument("--project", "-pj", type=str)
    p.add_argument("--epoch", "-e", type=int, default=config.epoch)
    p.add_argument("--learning_rate", "-lr", type=float, default=config.learning_rate)
    p.add_argument("--lr_sched", "-sch", action="store_true")
    p.add_argument("--perturb_budget", "-pb", type=float, default=config.perturb_budget)
    p.add_argument("--output_dir", "-od", type=str)
    p.add_argument("--tag", "-t", type=str)
    p.add_argument("--gpu", "-g", type=str, default=config.gpu)
    p.add_argument("--ckpt", "-c", type=str, required=True)
    p.add_argument("--batch", "-b", type=int, default=config.batch)
    p.add_argument("--dataset_path", "-dp", type=str, default=config.dataset_path)
    for k, v in vars(p.parse_args(argv)).items():
        if v is not None:
            setattr(config, k, v)
    config.output_dir = os.path.join(config.output_dir or "", naming_fn(config))
    with open(os.path.join(config.ckpt, "args.json")) as f:      # the backdoor run's arguments name the trigger / target to defend against
        a = json.load(f)
    config.trigger, config.target, config.poison_rate, config.dataset = a["trigger"], a["target"], a["poison_rate"], a["dataset"]
    os.makedirs(config.output_dir, exist_ok=True)
    with open(os.path.join(config.output_dir, "config.json"), "w") as f:
        json.dump({k: v for k, v in config.__dict__.items()}, f, indent=2, default=str)
    return config


def get_data_loader(config, device="cuda"):          # anp_util.py:149-157: every row poisoned (clean_rate 0, poison_rate 1): batches carry image + backdoor pair
    root = config.dataset_path if config.dataset_path and os.path.isdir(str(config.dataset_path)) else None
    dsl = DatasetLoader(root=root, name=config.dataset, batch_size=config.batch, seed=config.seed, device=device, num_images=config.num_images)
    return dsl.set_poison(trigger_type=config.trigger, target_type=config.target, clean_rate=0, poison_rate=1).prepare_dataset(mode=DatasetLoader.MODE_FIXED)


def cosine_lr(step, warmup, total):                  # diffusers optimization.py:134-138 (get_cosine_schedule_with_warmup)
    if step < warmup:
        return float(step) / float(max(1, warmup))
    progress = float(step - warmup) / float(max(1, total - warmup))
    return max(0.0, 0.5 * (1.0 + math.cos(math.pi * progress)))


def save_grid(images, path):
    """images [n, H, W, C] float in [0, 1] -> one PNG grid (anp_util.py:45-58, 159-176)."""
    from PIL import Image
    n = len(images)
    r = int(math.floor(math.sqrt(n)))
    while n % r:
        r -= 1
    c = n // r
    u8 = (np.asarray(images) * 255).round().astype("uint8")
    h, w = u8.shape[1:3]
    grid = Image.new("RGB", size=(c * w, r * h))
    for i, im in enumerate(u8):
        grid.paste(Image.fromarray(np.squeeze(im)).convert("RGB"), box=(i % c * w, i // c * h))
    grid.save(path)


def sampling(config, file_name, pipeline):           # anp_util.py:159-215 (clean noise only)
    d = os.path.join(config.output_dir, "samples")
    os.makedirs(d, exist_ok=True)
    s = pipeline.unet.sample_size
    noise = torch.randn((config.eval_sample_n, pipeline.unet.in_channels, s, s), generator=torch.manual_seed(config.seed))
    res = pipeline(batch_size=config.eval_sample_n, generator=torch.manual_seed(config.seed), init=noise, output_type=None)
    tag = f"{file_name:04d}" if isinstance(file_name, int) else str(file_name)
    save_grid(res.images, os.path.join(d, f"{tag}{'' if config.clip else '_noclip'}.png"))


def update_score_file(config, mse_sc, ssim_sc, epoch):     # anp_util.py:217-260 (per-epoch lists)
    path = os.path.join(config.output_dir, config.score_file)
    sc = json.load(open(path)) if os.path.exists(path) else {}
    sc.setdefault("epoch", []).append(epoch); sc.setdefault("MSE", []).append(mse_sc); sc.setdefault("SSIM", []).append(ssim_sc)
    with open(path, "w") as f:
        json.dump(sc, f, indent=2)
    return sc


def measure(config, pipeline, dsl, epoch=None):     # anp_defense.py:77-111: samples from CLEAN noise scored against the backdoor target
    epoch = epoch + 1 if epoch is not None else config.epoch
    s = pipeline.unet.sample_size
    noise = torch.randn((config.measure_sample_n, pipeline.unet.in_channels, s, s), generator=torch.manual_seed(config.seed))
    rng = torch.Generator(); rng.manual_seed(config.seed)
    imgs = batch_sampling(sample_n=config.measure_sample_n, pipeline=pipeline, init=noise, max_batch_n=config.eval_max_batch, rng=rng)
    gen = torch.from_numpy(np.asarray(imgs)).to(pipeline.unet.device).permute(0, 3, 1, 2).contiguous().float()
    gen = (gen * 255).round() / 255                                      # what ImagePathDataset reads back from the PNGs (anp_defense.py:94-96)
    tgt = (dsl.target.to(gen.device).float() / 2 + 0.5).clamp(0, 1).unsqueeze(0).expand_as(gen).contiguous()
    mse_sc = float(ops.loss_fwd_bwd(gen.permute(0, 2, 3, 1).contiguous(), tgt.permute(0, 2, 3, 1).contiguous(), "l2", want_grad=False)[0])
    ssim_sc = float(ops.ssim(gen, tgt, data_range=1.0))
    print(f"[{epoch}] MSE: {mse_sc}, SSIM: {ssim_sc}")
    update_score_file(config, mse_sc, ssim_sc, epoch)
    return mse_sc, ssim_sc


def train_loop(config, model, noise_sched, dsl, log=print):      # anp_defense.py:113-186
    trainer = anp.AnpTrainer(model, noise_sched, anp.AnpConfig(learning_rate=config.learning_rate, perturb_budget=config.perturb_budget,
                                                              epoch=config.epoch, batch=config.batch))
    g = torch.Generator(device=model.device); g.manual_seed(config.seed)
    cur_step, total = 0, dsl.num_batch * config.epoch
    history = []
    for epoch in range(int(config.epoch)):
        for batch in dsl.get_dataloader():
            clean, trig, targ = batch["image"], batch["pixel_values"], batch["target"]
            noise = torch.randn(trig.shape, device=trig.device, generator=g)
            t = torch.randint(0, noise_sched.num_train_timesteps, (trig.shape[0],), device=trig.device, generator=g).long()
            lr = config.learning_rate * cosine_lr(cur_step, config.lr_warmup_steps, total) if config.lr_sched else config.learning_rate
            logs = trainer.step(clean, trig, targ, t, noise, lr=lr)
            history.append({"loss": float(logs["loss"]), "backdoor_mse": float(logs["backdoor_mse"]), "clean_mse": float(logs["clean_mse"]),
                            "epoch": epoch, "step": cur_step, "lr": lr})
            cur_step += 1
        log(f"epoch {epoch}: " + json.dumps(history[-1]))
        pipeline = DDPMPipeline(unet=model, scheduler=noise_sched)
        if (epoch + 1) % config.save_image_epochs == 0:
            sampling(config, epoch, pipeline)
            measure(config, pipeline, dsl, epoch=epoch)
    pipeline = DDPMPipeline(unet=model, scheduler=noise_sched)
    sampling(config, "final", pipeline)
    measure(config, pipeline, dsl, epoch=None)
    torch.save({k: v.cpu() for k, v in model.state_dict().items() if ".bn." in k}, os.path.join(config.output_dir, "anp_bn.pt"))
    return pipeline, history


def main(argv=None):
    config = get_config(argv)
    dsl = get_data_loader(config)
    model, noise_sched, _ = DiffuserModelSched.get_pretrained(ckpt=config.ckpt, clip_sample=config.clip)      # anp_util.py:124
    perturb_model = anp.convert_model(model.cuda())                                                              # freeze + convert_model, anp_util.py:125-126
    return train_loop(config, perturb_model, noise_sched, dsl)


if __name__ == "__main__":
    main()

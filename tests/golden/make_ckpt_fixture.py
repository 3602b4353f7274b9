"""Capture what the REFERENCE's save_pretrained writes (G9), by importing the reference in the build container:

    python tests/golden/make_ckpt_fixture.py          # -> tests/golden/ckpt/

A DDPMPipeline(UNet2DModel(<small config>), DDPMScheduler(...)) is saved with the reference's own
pipeline_utils.py:527-600 / modeling_utils.py:287-301 and the resulting directory is reduced to DATA:
  * the JSON files verbatim (model_index.json, unet/config.json, scheduler/scheduler_config.json),
  * manifest.json: the file tree + for unet/diffusion_pytorch_model.bin every key, shape, dtype and contiguity,
  * weights_probe.npz: for every key the first 4 values and the fp64 sum of the tensor the reference stored -- the
    weights themselves are oracle.unet_ref.gen_params(cfg, seed), which the test regenerates.
No reference source text is stored.  The test (tests/test_host.py::test_reference_checkpoint_layout) checks that the
product loads this layout (after re-materialising the .bin from the regenerated weights) and writes one with identical
file names, JSON keys and tensor manifest."""
import json, os, sys, tempfile
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as MG            # noqa: E402  (import shim + reference imports; its g*() functions are not run)
import numpy as np
import torch

from oracle import unet_ref as U
from tests.golden import cases as C

SEED = 7


def main():
    cfg = C.SMALL_CFGS["small"]
    P = U.gen_params(cfg, SEED)
    unet = MG.UNet2DModel(sample_size=cfg.sample_size, in_channels=cfg.in_channels, out_channels=cfg.out_channels,
                          block_out_channels=cfg.block_out_channels, down_block_types=cfg.down_block_types,
                          up_block_types=cfg.up_block_types, layers_per_block=cfg.layers_per_block,
                          downsample_padding=cfg.downsample_padding, flip_sin_to_cos=cfg.flip_sin_to_cos, freq_shift=cfg.freq_shift,
                          norm_eps=cfg.norm_eps, norm_num_groups=cfg.norm_num_groups, attention_head_dim=cfg.attention_head_dim)
    unet.load_state_dict(P)
    sched = MG.DDPMScheduler(num_train_timesteps=1000, beta_start=0.0001, beta_end=0.02, clip_sample=False, variance_type="fixed_large")
    pipe = MG.DDPMPipeline(unet=unet, scheduler=sched)
    out = os.path.join(HERE, "ckpt")
    os.makedirs(out, exist_ok=True)
    with tempfile.TemporaryDirectory() as d:
        pipe.save_pretrained(d)
        tree = sorted(os.path.relpath(os.path.join(r, f), d) for r, _, fs in os.walk(d) for f in fs)
        for rel in tree:
            if rel.endswith(".json"):
                os.makedirs(os.path.dirname(os.path.join(out, rel)) or out, exist_ok=True)
                json.dump(json.load(open(os.path.join(d, rel))), open(os.path.join(out, rel), "w"), indent=2, sort_keys=True)
        sd = torch.load(os.path.join(d, "unet", "diffusion_pytorch_model.bin"), map_location="cpu")
        manifest = {"tree": tree, "seed": SEED, "config": "small",
                    "state_dict": [{"key": k, "shape": list(v.shape), "dtype": str(v.dtype), "contiguous": bool(v.is_contiguous())}
                                   for k, v in sd.items()]}
        json.dump(manifest, open(os.path.join(out, "manifest.json"), "w"), indent=1)
        np.savez_compressed(os.path.join(out, "weights_probe.npz"),
                            **{k: np.concatenate([v.flatten()[:4].double().numpy(), [float(v.double().sum())]]) for k, v in sd.items()})
    print("wrote", out, len(manifest["state_dict"]), "tensors;", tree)


if __name__ == "__main__":
    main()

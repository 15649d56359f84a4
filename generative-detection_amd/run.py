"""Thin runner for the hot path with the reference launcher's config surface (train.py:18-55,134-148,356-392,433):
`-b/--base` yaml files merged left to right, trailing `key=value` dotlist overrides, `-s/--seed` (default 23), the
learning-rate rule, then the PL-1.9-semantics loop of trainer.py on synthetic batches (the nuScenes pipeline, loggers,
checkpoint callbacks and the rest of train.py are out of scope).

    python -m odvae_amd.run -b tests/golden/autoencoder_kl_16x16x16.yaml --steps 4 --height 256 \
        model.params.lossconfig.params.disc_start=0 data.params.batch_size=8
"""
import argparse
import os
import random
import sys

import numpy as np
import torch

from . import synthetic
from .config import Config, configure_learning_rate, instantiate_from_config
from .trainer import Trainer


def seed_everything(seed):
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    os.environ["PL_GLOBAL_SEED"] = str(seed)


def parse(argv):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("-b", "--base", nargs="*", default=[], metavar="base_config.yaml")
    ap.add_argument("-s", "--seed", type=int, default=23)
    ap.add_argument("--scale_lr", type=lambda v: str(v).lower() in ("1", "true", "yes"), default=True)
    ap.add_argument("--steps", type=int, default=2, help="batches to run")
    ap.add_argument("--height", type=int, default=256, help="synthetic crop size (the yaml's patch_height)")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("-l", "--logdir", default=None, help="if given, the yaml's lightning.callbacks run and ImageLogger writes below it")
    return ap.parse_known_args(argv)


def main(argv=None):
    opt, unknown = parse(sys.argv[1:] if argv is None else argv)
    seed_everything(opt.seed)
    config = Config.merge(*[Config.load(p) for p in opt.base], Config.from_dotlist(unknown))
    lightning = config.pop("lightning", Config.create())
    trainer_cfg = lightning.get("trainer", Config.create())
    lp = config.model.params.lossconfig.params
    if "dataset_stats" not in lp and not os.path.exists(lp.get("dataset_stats_path", "dataset_stats/combined/all.pkl")):
        lp["dataset_stats"] = synthetic.dataset_stats_standin()   # the reference does not ship its stats pickle
    latent = opt.height // 2 ** (len(config.model.params.ddconfig.ch_mult) - 1)
    if latent != config.model.params.pose_decoder_config.params.n:
        for key in ("pose_decoder_config", "pose_encoder_config"):
            config.model.params[key].params["n"] = latent
            config.model.params[key].params["m"] = latent
        config.model.params["feat_dims"] = [config.model.params.embed_dim, latent, latent]
    model = instantiate_from_config(config.model)
    configure_learning_rate(config, model, trainer_cfg, scale_lr=opt.scale_lr, ngpu=1)
    model = model.to(opt.device).train()
    callbacks, logger = [], None
    if opt.logdir:   # the callbacks section of the yaml (train.py:452-463 instantiates them the same way)
        class _Logger:
            save_dir = opt.logdir
        logger = _Logger()
        for name, cb_cfg in lightning.get("callbacks", Config.create()).items():
            callbacks.append(instantiate_from_config(cb_cfg))
    trainer = Trainer(model, gradient_clip_val=trainer_cfg.get("gradient_clip_val", None), precision=trainer_cfg.get("precision", None),
                      callbacks=callbacks, logger=logger)
    bs = config.data.params.batch_size
    for step in range(opt.steps):
        batch = synthetic.make_batch(bs, opt.height, seed=opt.seed + step)
        losses = trainer.training_batch(batch, step)
        print("batch %d  global_step %d  aeloss %.4f  discloss %.4f  lr %.2e" %
              (step, model.global_step, losses[0].item(), losses[1].item(), model.learning_rate), flush=True)
    return model


if __name__ == "__main__":
    main()

"""Find host synchronisations inside a training step: torch.cuda.set_sync_debug_mode("warn") around a few steps.
usage: python tools/sync_probe.py [--gan] [--bf16]"""
import os, sys, warnings
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import synthetic
from odvae_amd.trainer import Trainer
YAML = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "autoencoder_kl_16x16x16.yaml")
dev = torch.device("cuda:0")
gan = "--gan" in sys.argv
kw = dict(perceptual_weight=1.0, disc_factor=1.0, disc_start=0) if gan else {}
torch.manual_seed(23)
model = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=4, ch=32, **kw).to(dev).train()
model._global_step = 1
trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0, 1) if gan else (0,), precision="bf16" if "--bf16" in sys.argv else None)
batch = synthetic.make_batch(4, 64, seed=23)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
def step(i):
    b = dict(batch); b["pose_6d"] = batch["pose_6d"].clone()
    return trainer.training_batch(b, i)
for i in range(3): step(i)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings(record=True) as w:
    warnings.simplefilter("always")
    step(3)
torch.cuda.set_sync_debug_mode("default")
print("synchronising calls in one step: %d" % len(w))
import collections
c = collections.Counter("%s:%d" % (os.path.basename(x.filename), x.lineno) for x in w)
for k, v in c.most_common(20): print("  %3d  %s" % (v, k))

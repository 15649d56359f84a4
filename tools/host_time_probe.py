"""Host time per training step (python + launch calls, no synchronisation) against the GPU time per step: how much slack the
host has.  usage: python tools/host_time_probe.py [--bf16]"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import synthetic
from odvae_amd.trainer import Trainer
YAML = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "autoencoder_kl_16x16x16.yaml")
dev = torch.device("cuda:0")
torch.manual_seed(23)
model = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=(4 if "--tiny" in sys.argv else 16)).to(dev)
model.train(); model._global_step = 1
if "--bf16" in sys.argv:
    model.set_precision("bf16")
trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0,))
B, R = (1, 64) if "--tiny" in sys.argv else (32, 256)     # --tiny: the GPU work vanishes, what is left is the host's own time
batch = synthetic.make_batch(B, R, seed=23)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
def step(i):
    b = dict(batch); b["pose_6d"] = batch["pose_6d"].clone()
    return trainer.training_batch(b, i)
for i in range(5): step(i)
torch.cuda.synchronize()
n = 10
t0 = time.perf_counter()
for i in range(n): step(5 + i)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("host %.1f ms/step to enqueue, %.1f ms/step until the GPU is done" % ((t1 - t0) / n * 1e3, (t2 - t0) / n * 1e3))

"""Which torch (aten) ops still launch kernels in one fp32 training step, by count and by source line (torch.profiler, with_stack)."""
import os, sys, collections, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from odvae_amd import synthetic
from odvae_amd.trainer import Trainer
YAML = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "autoencoder_kl_16x16x16.yaml")
dev = torch.device("cuda:0")
torch.manual_seed(23)
model = synthetic.build_model(YAML, batch_size_for_lr=12, latent_hw=16).to(dev)
model.train(); model._global_step = 1
trainer = Trainer(model, gradient_clip_val=1.0, optimizer_indices=(0,))
batch = synthetic.make_batch(8, 256, seed=23)
batch = {k: (v.to(dev) if torch.is_tensor(v) else v) for k, v in batch.items()}
def step(i):
    b = dict(batch); b["pose_6d"] = batch["pose_6d"].clone()
    return trainer.training_batch(b, i)
for i in range(2): step(i)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step(2); torch.cuda.synchronize()
by_op = collections.Counter(); by_line = collections.Counter()
for ev in prof.events():
    if ev.device_type == torch.autograd.DeviceType.CPU and ev.name.startswith("aten::") and ev.kernels if hasattr(ev, "kernels") else False:
        by_op[ev.name] += len(ev.kernels)
        src = next((f for f in ev.stack if "generative-detection_amd" in f or "odvae_amd" in f), ev.stack[0] if ev.stack else "?")
        by_line[(ev.name, src.strip()[:110])] += len(ev.kernels)
print("kernel launches by aten op:")
for k, v in by_op.most_common(25): print("  %4d  %s" % (v, k))
print("by op and first package frame:")
for k, v in by_line.most_common(40): print("  %4d  %-28s %s" % (v, k[0], k[1]))

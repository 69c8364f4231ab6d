import sys, json
sys.path.insert(0, '/root/repo')
import torch, bench
from inpaintnet_amd import synthetic
from inpaintnet_amd.measure_vae import MeasureVAE
ds = synthetic.SyntheticFolkDataset(num_notes=bench.NUM_NOTES)
model = MeasureVAE(ds)
print(json.dumps(bench.latent_rnn_extra(ds, model, torch.device('cuda:0'))))

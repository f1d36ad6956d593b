import time, threading, numpy as np, torch
from apr_amd import synth
from apr_amd.predator.configs.models import kitti_config
from apr_amd.predator.models.architectures import KPFCNN
from apr_amd.predator.pipeline import PredatorRegistration
dev = torch.device('cuda:0')
cfg = kitti_config()
torch.manual_seed(0)
model = KPFCNN(cfg).to(dev).eval()
pipe = PredatorRegistration(model, cfg, [58, 59, 58, 57])
pairs = []
for s in range(12):
    a, b, _ = synth.make_pair(300 + s, n_beams=64, n_azimuth=1800)
    pairs.append((torch.from_numpy(a).to(dev), torch.from_numpy(b).to(dev)))
single = [pipe(*p, seed=i) for i, p in enumerate(pairs[:4])]
got = pipe.register_batch(pairs[:4], seeds=range(4))
for (Ta, ia), (Tb, ib) in zip(single, got):
    print("pose diff", float(np.abs(Ta - Tb).max()), ia.get("inliers"), ib.get("inliers"))
STREAMS = [torch.cuda.Stream(device=dev) for _ in range(4)]
def run(T, B, total=48):
    nb = total // B
    batches = [[pairs[(i * B + j) % 12] for j in range(B)] for i in range(nb)]
    def work(w, warm):
        torch.cuda.set_device(dev)
        with torch.cuda.stream(STREAMS[w]):
            for i in (range(nb) if warm else range(w, nb, T)):
                pipe.register_batch(batches[i], seeds=range(B))
            STREAMS[w].synchronize()
    for rep in range(3):
        torch.cuda.synchronize(); t = time.perf_counter()
        ts = [threading.Thread(target=work, args=(w, rep == 0)) for w in range(T)]
        [x.start() for x in ts]; [x.join() for x in ts]
        torch.cuda.synchronize(); dt = time.perf_counter() - t
    print(f"threads {T} stacked {B}: {nb * B / dt:7.1f} pairs/s  ({dt / (nb * B) * 1e3:.2f} ms/pair)", flush=True)
for T, B in ((1, 1), (2, 1), (3, 1), (1, 2), (1, 4), (2, 2), (2, 4), (3, 2), (3, 4), (4, 4), (2, 6), (3, 6)):
    run(T, B)

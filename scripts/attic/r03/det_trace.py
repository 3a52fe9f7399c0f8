import os, sys, time, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import vsr_amd
from vsr_amd.synth import make_det_weights
from vsr_amd.backend.tools import ocr_det
from vsr_amd.backend.tools.paddle_graph import load_graph
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
g = load_graph(os.path.join(ROOT, "tests", "golden", "ppocr_det_graph.json"))
det = ocr_det.TextDetection(g, make_det_weights(g), device=0)
img = np.random.default_rng(3).integers(0, 256, size=(1080, 1920, 3), dtype=np.uint8)
imgs = [img] * 8
for _ in range(3): det.probability_maps(imgs)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): det.probability_maps(imgs)
torch.cuda.synchronize(); print("8 frames per forward: %.2f ms/frame" % ((time.perf_counter() - t0) / 5 / 8 * 1e3))

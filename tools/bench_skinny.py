import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sonicscribe_amd import spec
from sonicscribe_amd.engine import Engine
eng = Engine(spec.TINY, 0, max_batch=2, max_ctx=128)
eng.load_synthetic(1)
names = {0: "xs/dma-nt", 3: "reg", 9: "readfloor"}
for (M, N, K) in [(32, 12288, 2048), (32, 2048, 6144), (32, 3072, 2048), (32, 2048, 2048), (32, 59264, 2048), (64, 12288, 2048), (16, 12288, 2048)]:
    mb = N * K * 2 / 1e6
    line = f"M={M} N={N} K={K} ({mb:.1f} MB): "
    for rnd in range(2):
        for v in (0, 3, 9):
            us = eng.bench_skinny(M, N, K, v, 40)
            if rnd == 1:
                line += f"{names[v]} {us:.1f}us ({mb / us / 1e3:.2f} TB/s)  "
    print(line)

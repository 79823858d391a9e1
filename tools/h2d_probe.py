"""Raw H2D bandwidth of this box (the ceiling of slideo_match_frames_bgr8's PCIe-inclusive rate): one 1.6 GB copy from
pageable / pinned memory, whole and in 32-frame pieces on 1 / 2 / 4 streams."""
import time, torch
n = 256 * 1920 * 1080 * 3
src = torch.empty(n, dtype=torch.uint8).random_(0, 255)
pin = src.pin_memory()
dst = torch.empty(n, dtype=torch.uint8, device="cuda")
def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps
for name, s in (("pageable", src), ("pinned", pin)):
    dt = t(lambda: dst.copy_(s, non_blocking=True))
    print("%s whole: %.1f ms  %.1f GB/s" % (name, dt * 1e3, n / dt / 1e9))
    for ns in (1, 2, 4):
        streams = [torch.cuda.Stream() for _ in range(ns)]
        piece = n // 8
        def go():
            for i in range(8):
                with torch.cuda.stream(streams[i % ns]):
                    dst[i * piece:(i + 1) * piece].copy_(s[i * piece:(i + 1) * piece], non_blocking=True)
        dt = t(go)
        print("%s 8 pieces on %d stream(s): %.1f ms  %.1f GB/s" % (name, ns, dt * 1e3, n / dt / 1e9))

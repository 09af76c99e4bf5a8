"""What plain streaming passes reach on this GPU (context for the `hbm` roofline fractions of bench.py, which are priced against the
8 TB/s of MI355X_MICROARCH.md): fill (write only), sum (read only), copy (read + write) of a 1 GiB float32 tensor with the
library kernels PyTorch ships, 20 repetitions between one pair of HIP events."""
import torch


def timed(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e-3


def main():
    n = 1 << 28
    x = torch.ones(n, dtype=torch.float32, device="cuda")
    y = torch.empty_like(x)
    gb = n * 4 / 1e9
    print(f"fill  (write {gb:.2f} GB)          {gb / timed(lambda: y.fill_(2.0)) / 1e3:6.2f} TB/s")
    print(f"sum   (read {gb:.2f} GB)           {gb / timed(lambda: x.sum()) / 1e3:6.2f} TB/s")
    print(f"copy  (read + write {2 * gb:.2f} GB)   {2 * gb / timed(lambda: y.copy_(x)) / 1e3:6.2f} TB/s")
    c = torch.ones(n // 2, dtype=torch.complex64, device="cuda")
    d = torch.empty_like(c)
    e = torch.ones_like(c)
    print(f"mul   (2 reads + 1 write {3 * gb:.2f} GB) {3 * gb / timed(lambda: torch.mul(c, e, out=d)) / 1e3:6.2f} TB/s")


if __name__ == "__main__":
    main()

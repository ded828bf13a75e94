"""Same-node yardsticks for the dense solve (TEST / PROFILE ONLY -- nothing here is used by the product):
rocSOLVER dpotrf + dpotrs through torch.linalg at the order of the reduced camera system, and the f64 GEMM rate rocBLAS
sustains on this GPU (what the matrix cores deliver to a vendor kernel).  Prints one JSON line."""
import json
import sys
import time

import torch


def timeit(fn, reps=5):
    fn(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return best


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 6016
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(1)
    B = torch.randn((n, n), dtype=torch.float64, device=dev, generator=g)
    A = B @ B.T + n * torch.eye(n, dtype=torch.float64, device=dev)
    b = torch.randn((n, 1), dtype=torch.float64, device=dev, generator=g)
    out = {"n": n, "device": torch.cuda.get_device_name(0), "torch": torch.__version__}
    t = timeit(lambda: torch.linalg.cholesky(A))
    out["potrf_ms"] = 1e3 * t; out["potrf_tflops"] = n ** 3 / 3 / t / 1e12
    L = torch.linalg.cholesky(A)
    t2 = timeit(lambda: torch.cholesky_solve(b, L))
    out["potrs_ms"] = 1e3 * t2
    t3 = timeit(lambda: torch.cholesky_solve(b, torch.linalg.cholesky(A)))
    out["potrf_plus_potrs_ms"] = 1e3 * t3
    for m in (2048, 4096, 6016, 8192):
        X = torch.randn((m, m), dtype=torch.float64, device=dev, generator=g)
        Y = torch.randn((m, m), dtype=torch.float64, device=dev, generator=g)
        tg = timeit(lambda: X @ Y, reps=8)
        out[f"dgemm_{m}_tflops"] = 2 * m ** 3 / tg / 1e12
    # rank-128 / rank-256 updates of a 6016 matrix: the shape of the factorisation's trailing update
    C = torch.zeros((n, n), dtype=torch.float64, device=dev)
    for k in (128, 256, 512):
        P = torch.randn((n, k), dtype=torch.float64, device=dev, generator=g)
        tk = timeit(lambda: torch.addmm(C, P, P.T, beta=1.0, alpha=-1.0, out=C), reps=8)
        out[f"rank{k}_update_full_square_tflops"] = 2 * n * n * k / tk / 1e12
    print(json.dumps(out))


if __name__ == "__main__":
    main()

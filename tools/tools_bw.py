import torch, time
dev = torch.device("cuda", 0)
def timeit(fn, n=20):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
N = 4096 * 4096
x = torch.rand(16 * N, device=dev)
y = torch.empty_like(x)
t = timeit(lambda: y.copy_(x))
print(f"copy 1 GiB: {2*x.numel()*4/t/1e12:.2f} TB/s")
# 5 reads + 4 writes, arrays exactly 64 MiB apart (one allocation) vs padded
for pad in [0, 4096 + 64]:
    stride = N + pad
    buf = torch.rand(9 * stride, device=dev)
    a = [buf[i * stride:i * stride + N] for i in range(9)]
    def f():
        torch._foreach_add_(a[5:9], a[0:4])   # 4 x (2 reads + 1 write)
    t = timeit(f)
    print(f"pad {pad}: foreach 4x(2r+1w): {12*N*4/t/1e12:.2f} TB/s")
    def g():
        torch.add(a[0], a[1], out=a[5])
    t = timeit(g)
    print(f"pad {pad}: add 2r+1w 64MiB arrays: {3*N*4/t/1e12:.2f} TB/s")
# read-only and write-only streams (is the ceiling a property of reads, of writes, or of the mix?)
t = timeit(lambda: x.sum())
print(f"sum (read only) 1 GiB: {x.numel()*4/t/1e12:.2f} TB/s")
t = timeit(lambda: y.fill_(1.0))
print(f"fill (write only) 1 GiB: {y.numel()*4/t/1e12:.2f} TB/s")
h = x.numel() // 2
t = timeit(lambda: torch.add(x[:h], x[h:], out=y[:h]))
print(f"add 2r+1w 512 MiB arrays: {3*h*4/t/1e12:.2f} TB/s")

"""Developer tool: run the vendor GEMM once per shape so that `rocprofv3 --kernel-trace --stats`
shows which library kernels (names encode macro-tile / wave layout) serve the BERT-large shapes."""
import torch
M = 32768
for n, k in [(3072, 1024), (1024, 1024), (4096, 1024), (1024, 4096)]:
    x = torch.randn(M, k, device='cuda').to(torch.bfloat16)
    w = (torch.randn(n, k, device='cuda') * 0.02).to(torch.bfloat16)
    for _ in range(5):
        y = torch.matmul(x, w.t())
    torch.cuda.synchronize()

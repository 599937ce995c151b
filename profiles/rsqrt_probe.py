import torch
g = torch.Generator(device="cuda").manual_seed(0)
v = (torch.rand(1 << 22, device="cuda", generator=g) * 50 + 1e-3)
eps = 1e-5
a = torch.rsqrt(v + eps)
b = 1.0 / torch.sqrt(v + eps)
c = torch.sqrt(1.0 / (v + eps))
d = (v + eps).double().rsqrt().float()   # correctly rounded rsqrt
print("rsqrt vs 1/sqrt mismatches:", int((a != b).sum()), " vs sqrt(1/x):", int((a != c).sum()), " vs correctly rounded:", int((a != d).sum()),
      " 1/sqrt vs correctly rounded:", int((b != d).sum()))

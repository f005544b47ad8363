import torch
f, t = torch.cuda.mem_get_info()
print("PPD free %.2f GiB total %.2f GiB" % (f / 2**30, t / 2**30))

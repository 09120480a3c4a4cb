import sys, json, torch
sys.path.insert(0,'/root/repo')
from libcimbar_amd import HipDecoder, framegen
from tools import extractbench
dev=torch.device("cuda",0)
dec=HipDecoder(0)
print(json.dumps(extractbench.run(dec, dev, torch.cuda.current_stream(dev), None)))

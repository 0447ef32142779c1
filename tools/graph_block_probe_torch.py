"""tools/graph_block_probe.hip inside a process that has torch's HIP runtime initialised (is the host wait of a graph launch a
property of the process?)"""
import ctypes, os, sys
import torch
torch.cuda.init(); x = torch.zeros(1024, device="cuda"); torch.cuda.synchronize()
if len(sys.argv) > 1 and sys.argv[1] == "streams":
    keep = [torch.cuda.Stream() for _ in range(3)]
lib = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph_block_probe_lib.so"))
n = sys.argv[2].encode() if len(sys.argv) > 2 else b"200"
d = sys.argv[3].encode() if len(sys.argv) > 3 else b"8"
argv = (ctypes.c_char_p * 3)(b"probe", n, d)
lib.probe_main(3, argv)

"""A/B: the gradient sink's weight gradients on the MAIN stream (no overlap) instead of its side stream.  Usage: as bench.py."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import ideas_amd.op.conv as CV
CV._SINK["stream"] = torch.cuda.current_stream()
import bench
bench.main()

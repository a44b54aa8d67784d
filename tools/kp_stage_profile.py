#!/usr/bin/env python3
"""Development aid: the keypoint stage of a 32-pair step alone on the chip, five times (for rocprofv3 --kernel-trace --stats)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dataclasses import replace
from buffer_amd import synth
from buffer_amd.config import THREEDMATCH
from buffer_amd.pipeline import BufferPipeline
dev = torch.device('cuda:0')
pipe = BufferPipeline(replace(THREEDMATCH, num_keypts=5000), dev)
pipe.calibrate([synth.make_pair(1000)])
inps = [pipe.upload(synth.make_pair(2000 + i)) for i in range(8)]
batch = [inps[k % 8] for k in range(32)]
for _ in range(6):
    pipe._keypoints(batch, list(range(32)), None)
torch.cuda.synchronize()

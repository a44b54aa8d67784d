import sys, time, torch
sys.path.insert(0, '/root/repo')
from dataclasses import replace
from buffer_amd import synth
from buffer_amd.config import THREEDMATCH
from buffer_amd.pipeline import BufferPipeline
dev = torch.device('cuda:0')
pipe = BufferPipeline(replace(THREEDMATCH, num_keypts=5000), dev)
pipe.calibrate([synth.make_pair(1000)])
inputs = [pipe.upload(synth.make_pair(2000 + i)) for i in range(4)]
batches = [[inputs[(i * 32 + j) % 4] for j in range(32)] for i in range(4)]
seeds = [[i * 32 + j for j in range(32)] for i in range(4)]
pipe.register_batches(batches[:2], seeds[:2]); torch.cuda.synchronize()
main = torch.cuda.current_stream(dev); side = pipe._kp_stream
T0 = time.perf_counter()
def now(): return round((time.perf_counter() - T0) * 1e3, 1)
def stage1(i):
    with torch.cuda.stream(side):
        st = pipe._keypoints(batches[i], seeds[i], None); ev = torch.cuda.Event(); ev.record(side)
    return st, ev
side.wait_stream(main)
t = now(); nxt = stage1(0); print('stage1(0) host', t, now())
for i in range(4):
    st, ev = nxt
    main.wait_event(ev)
    t = now(); st = pipe._describe(st); print(i, 'describe host', t, now())
    t = now(); nxt = stage1(i + 1) if i + 1 < 4 else None; print(i, 'stage1 next host', t, now())
    t = now(); pipe._match(st); print(i, 'match host', t, now())
torch.cuda.synchronize(); print('end', now())

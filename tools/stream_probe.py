"""Development probe: host-side timeline of the streamed workload (16-pair batches, 1500 keypoints, pre-processing on the
side stream): where the host thread spends a batch."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from buffer_amd import stream, synth
from buffer_amd.config import THREEDMATCH
from buffer_amd.pipeline import BufferPipeline
dev = torch.device('cuda:0')
pipe = BufferPipeline(THREEDMATCH, dev)
raws = stream.generate(96, dev)
pipe.calibrate([synth.make_pair(1000)])
B = 16
ids = [list(range(i, i + B)) for i in range(0, 96, B)]
upload = stream.upload
def maker(idl):
    return lambda: [upload(s) for s in stream.prepare_batch([raws[i] for i in idl], pipe.cfg, idl)]
stream.run(pipe, raws[:32], batch=B); torch.cuda.synchronize()
main = torch.cuda.current_stream(dev); side = pipe._kp_stream
T0 = time.perf_counter()
def now(): return (time.perf_counter() - T0) * 1e3
def stage1(idl):
    with torch.cuda.stream(side):
        t0 = now(); inps = maker(idl)(); t1 = now()
        st = pipe._keypoints(inps, idl, None); t2 = now()
        st['cross'] = tuple(st.get('cross', ())) + tuple(v for x in inps for v in x.values() if isinstance(v, torch.Tensor))
        ev = torch.cuda.Event(); ev.record(side)
    return st, ev, t1 - t0, t2 - t1
side.wait_stream(main)
nxt = stage1(ids[0])
for i in range(len(ids)):
    st, ev, tp, tk = nxt
    main.wait_event(ev)
    for t in st.get('cross', ()): t.record_stream(main)
    t0 = now(); st = pipe._describe(st); t1 = now()
    nxt = stage1(ids[i + 1]) if i + 1 < len(ids) else None; t2 = now()
    pipe._match(st); t3 = now()
    print(f'batch {i}: describe enqueue {t1 - t0:6.1f} ms | next prepare {nxt[2] if nxt else 0:6.1f} + keypoints {nxt[3] if nxt else 0:6.1f} ms | match (waits for the GPU) {t3 - t2:6.1f} ms | total {t3 - t0:6.1f}')
torch.cuda.synchronize(); print(f'end {now():.1f} ms for {len(ids) * B} pairs')

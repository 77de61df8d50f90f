"""What the host side of run_train(x, y) costs when the caller passes pageable host tensors and wants the image back on the host
(SISRInterface.train_batch, rumpy/SISR/models/interface.py:97-101): the pieces of HipSRNet._stage_out / BaseModel._to_device timed one by one."""
import time

import torch

dev = torch.device('cuda', 0)
N = 32
x, y = torch.rand(N, 3, 48, 48), torch.rand(N, 3, 192, 192)
out = torch.rand(N, 3, 192, 192, device=dev)
torch.cuda.synchronize()


def t(fn, reps=20, sync=True):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    if sync:
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print('pageable y.to(device) [14 MB]                 %.3f ms' % t(lambda: y.to(dev)))
print('out.cpu() [14 MB, pageable destination]        %.3f ms' % t(lambda: out.cpu()))
pin = torch.empty(y.shape, pin_memory=True)
print('pinned.copy_(pageable y) [host memcpy]         %.3f ms' % t(lambda: pin.copy_(y)))
print('pinned -> device, non_blocking                 %.3f ms' % t(lambda: pin.to(dev, non_blocking=True)))
print('torch.empty(pin_memory=True) + drop            %.3f ms' % t(lambda: torch.empty(y.shape, pin_memory=True)))
print('device -> fresh pinned, non_blocking + sync    %.3f ms' % t(lambda: torch.empty(y.shape, pin_memory=True).copy_(out, non_blocking=True)))
side = torch.cuda.Stream(dev)


def staged():
    host = torch.empty(out.shape, pin_memory=True)
    side.wait_stream(torch.cuda.current_stream(dev))
    with torch.cuda.stream(side):
        host.copy_(out, non_blocking=True)
        ev = torch.cuda.Event()
        ev.record(side)
    ev.synchronize()
    return host


print('_stage_out + wait (side stream, fresh pinned)  %.3f ms' % t(staged))
keep = [staged() for _ in range(4)]
print('... while 4 earlier results are still alive    %.3f ms' % t(staged))
torch.set_num_threads(8)
print('pinned.copy_(pageable y), 8 threads            %.3f ms' % t(lambda: pin.copy_(y)))

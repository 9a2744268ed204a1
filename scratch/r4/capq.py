# does hipEventQuery on an event recorded EAGERLY fail once the event's stream has joined a capture?
import threading, torch
dev = torch.device("cuda:0")
x = torch.zeros(1 << 20, device=dev)
S, C = torch.cuda.Stream(), torch.cuda.Stream()
E = torch.cuda.Event()
with torch.cuda.stream(S):
    x.add_(1.0)
    E.record(S)
torch.cuda.synchronize()
print("before capture: query ->", E.query())
res = {}
def poll(tag):
    try:
        res[tag] = E.query()
    except Exception as e:  # noqa
        res[tag] = "ERROR " + str(e).splitlines()[0]
g = torch.cuda.CUDAGraph()
try:
  with torch.cuda.stream(C):
    with torch.cuda.graph(g, stream=C, capture_error_mode="thread_local"):
        x.mul_(2.0)
        t = threading.Thread(target=poll, args=("other thread, S not yet in the capture",)); t.start(); t.join()
        fork = torch.cuda.Event(); fork.record(C)
        S.wait_event(fork)           # S joins the capture
        with torch.cuda.stream(S):
            x.add_(3.0)
        t = threading.Thread(target=poll, args=("other thread, S IN the capture",)); t.start(); t.join()
        join = torch.cuda.Event(); join.record(S)
        C.wait_event(join)
except Exception as e:
    print("capture:", str(e).splitlines()[0])
for k, v in res.items():
    print(k, "->", v)
torch.cuda.synchronize()
print("after capture: query ->", E.query())

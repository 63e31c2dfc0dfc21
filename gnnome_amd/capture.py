"""hipGraph capture of a whole forward.

One `model(graph, x, e)` is ~30 kernel launches; on a small graph (or a slow host) the Python + launch cost of
issuing them (0.6-2.7 ms measured on the pool's hosts) exceeds the GPU time.  `CapturedForward` records the launch
sequence once into a hipGraph (torch.cuda.CUDAGraph over the library's plain stream-ordered launches) and replays
it with one call; inputs are copied into the static buffers the recording used.  Inference only.
"""
import torch

from .graph import views_for


class CapturedForward:
    def __init__(self, model, graph, x, e):
        if model.training:
            raise RuntimeError("CapturedForward is for eval-mode inference")
        device = x.device if x.is_cuda else torch.device("cuda", torch.cuda.current_device())
        self.model, self.device = model, device
        self.views = views_for(graph, device)
        self.x = x.detach().to(device=device, dtype=torch.float32).contiguous().clone()
        self.e = e.detach().to(device=device, dtype=torch.float32).contiguous().clone()
        side = torch.cuda.Stream(device=device)
        side.wait_stream(torch.cuda.current_stream(device))
        with torch.cuda.stream(side), torch.no_grad():   # warm-up: weight preparation, allocator pools
            for _ in range(2):
                model(self.views, self.x, self.e)
        torch.cuda.current_stream(device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.out = model(self.views, self.x, self.e)
        from .engine import _StateProbe
        self._probe = _StateProbe(model, device)   # the recording holds the weights as PREPARED at capture time

    def __call__(self, x=None, e=None):
        """Replay; new features (same shapes, same graph) may be supplied.  Returns the static output tensor.  A replay is NOT range-checked
        (engine.forward_in_range looked at the inputs the recording was made with): features beyond fp16x3's range (|x| >= 65504 somewhere in
        the stack) come back as NaN rows, never as wrong finite values - record under `ops.bf16x6_arithmetic()` for such inputs."""
        if not self._probe.unchanged(self.model, self.device):
            raise RuntimeError("the model's parameters or buffers changed since this forward was captured (the recording "
                               "replays the weights prepared at capture time): build a new CapturedForward")
        if x is not None:
            self.x.copy_(x)
        if e is not None:
            self.e.copy_(e)
        self.graph.replay()
        return self.out

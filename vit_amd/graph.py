"""HIP-graph replay of the forward for launch-bound batch sizes.

One forward is ~90 kernel launches; at batch 256 the GPU time per launch (>= 35 us) hides the host's ~5 us per
launch, at batch 1-8 it does not.  `GraphedForward` captures `model(x)` for one input shape into a HIP graph
(through torch.cuda.CUDAGraph: the kernels are launched on torch's current stream, so stream capture sees
them) and replays it with one host call.  The captured graph reads the kernel-ready weight copies that existed
at capture time: after changing the parameters, capture again.
"""
import torch

from .lib import VitkError


class GraphedForward:
    def __init__(self, model, example, warmup=3):
        if not example.is_cuda:
            raise VitkError("GraphedForward needs a CUDA/HIP example input")
        if model.training or torch.is_grad_enabled() and any(p.requires_grad for p in model.parameters()):
            # the autograd path allocates per op and keeps tensors alive for the backward: not a static graph
            if torch.is_grad_enabled():
                raise VitkError("capture the forward under torch.no_grad()")
        self.model = model
        self.static_in = example.clone()
        side = torch.cuda.Stream(device=example.device)
        side.wait_stream(torch.cuda.current_stream(example.device))
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                     # builds the weight cache and workspaces outside the capture
                model(self.static_in)
        torch.cuda.current_stream(example.device).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(self.graph):
            self.static_out = model(self.static_in)

    def __call__(self, x):
        """Logits for x (same shape and dtype as the example).  The returned tensor is overwritten by the next call."""
        if x.shape != self.static_in.shape or x.dtype != self.static_in.dtype:
            raise VitkError(f"graph was captured for {tuple(self.static_in.shape)} {self.static_in.dtype}, "
                            f"got {tuple(x.shape)} {x.dtype}")
        self.static_in.copy_(x, non_blocking=True)
        self.graph.replay()
        return self.static_out

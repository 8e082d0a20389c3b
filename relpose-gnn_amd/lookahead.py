"""The reference's evaluation loop AS IT IS WRITTEN, at micro-batched rates.

``MultiModelTrainer.eval_RP`` walks a ``DataLoader(batch_size=1)`` and calls ``self.model(data.to(self.device))`` once per
graph (/root/reference/python/niantic/testing/test.py:192-211), reading the poses back before the next graph is touched: on
an MI355X that is one 8-node forward -- a chain of ~130 launch-sized kernels, ~1.6 ms -- per iteration, a fifth of what the
same module sustains on 64 graphs per forward.  ``evaluate.evaluate_stream`` is the product's loop; this module is for a
caller who keeps the reference's loop body (its ``.cpu().data.numpy()`` post-processing, its ``tqdm``, its file bookkeeping,
test.py:213-286) and changes ONE line in front of it::

    loader, self.model = lookahead(loader, self.model, self.device, micro_batch=64)      # after test.py:193-194

The wrapped loader reads ``micro_batch`` graphs AHEAD of the loop, sends their node images through the pinned double
buffers of ``evaluate._InputPipeline``, runs ONE forward over them and copies the poses to pinned host memory -- while the
loop is still consuming the previous micro-batch -- and then yields the graphs one by one; the wrapped model hands each
``model(data.to(device))`` call that graph's rows of the batched result (results per graph do not depend on the batching:
``tests/test_hip_model.py::test_batch_independence_full_width``).  What the loop sees:

* ``data``: a shallow copy of the loader's item whose ``x`` already lives on the device (the rows of the staging buffer the
  forward read); ``data.to(device)`` returns it as it is -- ``y`` / ``edge_index`` stay where the loader put them, the loop only
  reads them back (test.py:216-219) --; ``len(data)``, ``data.y`` ... are the item's own;
* ``output, output_R, edge_index``: HOST tensors ([n, 6], [e, 6], [2, e] with the graph's own node ids) -- ``.cpu()``,
  ``.size()``, ``.data.numpy()`` work as on device tensors and cost nothing;
* ``len(loader)``, ``loader.batch_size``, ``loader.dataset``: the wrapped loader's.

A call with anything that was not the graph just yielded (a second forward, another loader's batch) runs the wrapped module
directly.  The reference's always-on dropout (posenet.py:1073-1075) draws its mask per micro-batch instead of per graph.
"""
from __future__ import annotations

import copy
from typing import Iterable, Optional

import numpy as np
import torch

from .evaluate import _MicroBatchRunner, edges_per_graph
from .graph import Data


class _Ticket:
    """What the wrapped loader yields: the loader's own item (a shallow copy whose ``x`` already lives on the device) behind a
    proxy whose ``to(device)`` is a no-op for the device the images are on.  The reference's loop calls ``data.to(self.device)``
    and then only READS BACK what it sent (``data.y.to('cpu')``, test.py:211-219): moving y / edge_index / batch over and back
    costs four small synchronous copies per graph, which on a busy GPU queue behind the batched forwards' resident workgroups and
    the next micro-batch's 537-MB image copy (measured round 6: 0.09 s or 0.7 s of loop body per 1024 graphs, by timing).  They
    stay where the loader put them; ``data.y.to(device)`` moves one explicitly.  Everything else is the item's."""
    __slots__ = ("_item",)

    def __init__(self, item):
        object.__setattr__(self, "_item", item)

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, "_item"), name)

    def __setattr__(self, name, value):
        setattr(object.__getattribute__(self, "_item"), name, value)

    def __len__(self):
        return len(object.__getattribute__(self, "_item"))

    def to(self, device, *args, **kwargs):
        item = object.__getattribute__(self, "_item")
        x = getattr(item, "x", None)
        d = torch.device(device) if not isinstance(device, torch.device) else device
        if torch.is_tensor(x) and x.device.type == d.type and (d.index is None or d.index == x.device.index):
            return self
        return item.to(device, *args, **kwargs)


class _AheadLoader:
    """Iterable stand-in for the caller's loader: same length / attributes, items come out ``micro_batch`` graphs behind
    the point the underlying loader has been read to."""

    def __init__(self, owner: "Lookahead", loader: Iterable):
        self._owner, self._loader = owner, loader

    def __iter__(self):
        return self._owner._run(iter(self._loader))

    def __len__(self):
        return len(self._loader)

    def __getattr__(self, name):            # batch_size, dataset, sampler, ... (only called for names not defined here)
        return getattr(self._loader, name)


class Lookahead:
    """The model side of ``lookahead()``: callable like the module, everything else (``eval()``, ``state_dict()``,
    ``index_check`` ...) is the wrapped module's."""

    def __init__(self, model, device, micro_batch: int = 64, bf16_input: Optional[bool] = None):
        if micro_batch < 1:
            raise ValueError("micro_batch must be >= 1")
        h2d = torch.bfloat16 if (bf16_input if bf16_input is not None else getattr(model, "accepts_bf16_input", False)) else torch.float32
        self.__dict__["_model"] = model
        self.__dict__["_runner"] = _MicroBatchRunner(model, device, micro_batch, h2d, want_abs=True, pinned_direct=bf16_input is None)
        self.__dict__["_expect"] = None      # (device pointer of the yielded graph's x, its (abs, rel, edge_index))
        # The batched forwards run on a stream of their own: the loop's own device traffic (data.to(device) sends y and
        # edge_index of every graph, test.py:211) is ordered on the CURRENT stream and would otherwise queue behind the forward
        # of the next micro-batch -- the first graph of every micro-batch waiting ~35 ms with the GPU idle afterwards (measured:
        # 1150 graphs/s on the current stream)
        self.__dict__["_stream"] = torch.cuda.Stream(device=device) if torch.device(device).type == "cuda" else None
        self.__dict__["forwards"] = 0        # batched forwards issued (tests / statistics)
        self.__dict__["direct_calls"] = 0    # calls that went to the module itself
        self.__dict__["_direct_pending"] = False   # a direct call ran on the caller's stream since the last batched launch
        self.__dict__["_in_loop"] = False    # the wrapped loader is being iterated
        self.__dict__["_warned"] = False

    def __getattr__(self, name):
        return getattr(self.__dict__["_model"], name)

    def __setattr__(self, name, value):      # model.index_check = "sync", model.encoder_dtype = ... reach the module
        if name in self.__dict__:
            self.__dict__[name] = value
        else:
            setattr(self.__dict__["_model"], name, value)

    def ahead(self, loader: Iterable) -> _AheadLoader:
        return _AheadLoader(self, loader)

    def __call__(self, data, k=None):
        exp = self._expect
        if exp is not None and k is None and torch.is_tensor(getattr(data, "x", None)) and data.x.data_ptr() == exp[0]:
            self.__dict__["_expect"] = None
            return exp[1]
        self.__dict__["direct_calls"] += 1
        if self._in_loop and exp is not None and k is None and not self._warned:
            # inside the wrapped loop, with a graph on offer, and the call did not present it: most likely the caller cloned /
            # re-laid-out data.x (the ticket is recognised by the device address of x).  Correct, but every such call is a
            # whole single-graph forward -- say so once instead of silently running at a third of the rate.
            import warnings
            warnings.warn("lookahead: model(data) inside the wrapped loop was not given the graph the loader just yielded (data.x is "
                          "a different tensor: cloned / reshaped into new memory?) -- running a separate forward for it; the batched "
                          "result of that graph is discarded", RuntimeWarning, stacklevel=2)
            self.__dict__["_warned"] = True
        if self._stream is not None:         # the module's workspaces are shared: a direct call waits for the forwards in flight
            torch.cuda.current_stream().wait_stream(self._stream)
            self.__dict__["_direct_pending"] = True     # ... and the next batched launch waits for this one (start())
        if isinstance(data, _Ticket):        # a ticket's to(device) left y / edge_index on the host: the module needs them next to x
            inner = object.__getattribute__(data, "_item")
            data = inner.to(inner.x.device) if torch.is_tensor(getattr(inner, "x", None)) else inner
        return self._model(data, k) if k is not None else self._model(data)

    # ---- the generator behind the wrapped loader ---------------------------------------------------------------
    @staticmethod
    def _as_graph(item) -> Data:
        x = item.x
        if x.dim() != 2:
            x = x.reshape(x.shape[0], -1)
        b = getattr(item, "batch", None)
        if getattr(item, "num_graphs", 1) != 1 or (torch.is_tensor(b) and b.numel() and int(b.max()) != 0):
            raise ValueError("lookahead() wraps the reference's batch_size = 1 loader (test.py:192): one graph per item")
        return Data(x=x, edge_index=item.edge_index, y=getattr(item, "y", None), edge_attr=None)

    @torch.no_grad()
    def _run(self, it):
        runner = self._runner
        mb = runner.micro_batch

        def start():
            items = []
            for item in it:
                items.append(item)
                if len(items) == mb:
                    break
            if not items:
                return None
            self.__dict__["forwards"] += 1
            graphs = [self._as_graph(d) for d in items]
            if self._stream is None:
                return items, runner.launch(graphs)
            if self._direct_pending:
                # a direct forward was enqueued on the caller's stream after the last batched launch: it uses the module's
                # default workspaces, which the next batched forward may share (hip_streams = 1, small micro-batches) -- order
                # the two in THIS direction too (the other direction is the wait in __call__)
                self._stream.wait_stream(torch.cuda.current_stream())
                self.__dict__["_direct_pending"] = False
            with torch.cuda.stream(self._stream):
                return items, runner.launch(graphs)

        if self._stream is not None:         # whatever the caller enqueued so far (weights, a direct forward) comes first
            self._stream.wait_stream(torch.cuda.current_stream())
            self.__dict__["_direct_pending"] = False
        self.__dict__["_in_loop"] = True
        try:
            yield from self._drain(start)
        finally:
            self.__dict__["_in_loop"] = False
            self.__dict__["_expect"] = None

    def _drain(self, start):
        cur = start()
        while cur is not None:
            nxt = start()                    # micro-batch i + 1 is staged and enqueued before micro-batch i is handed out
            items, (chunk, host_rel, host_ei, ev, host_abs, batch) = cur
            x_dev = batch.x
            if ev is not None:
                ev.synchronize()
            check = getattr(self._model, "check_edge_index", None)
            if check is not None:
                check(wait=False)            # raises the IndexError of a bad edge_index in THIS micro-batch (evaluate.finish)
            sizes = [g.num_nodes for g in chunk]
            cols = None
            if host_ei is not None:          # model-built edge list (kNN): cut at graph boundaries, local node ids
                ei = host_ei.numpy()
                first, cols = edges_per_graph(ei, sizes)
                rel_np = host_rel.numpy()
            n0 = e0 = 0
            for j, (item, g) in enumerate(zip(items, chunk)):
                n = sizes[j]
                if cols is None:
                    e = int(g.edge_index.shape[1])
                    rel_j, ei_j = host_rel[e0:e0 + e], g.edge_index if not g.edge_index.is_cuda else g.edge_index.cpu()
                    e0 += e
                else:
                    rel_j = torch.from_numpy(rel_np[cols[j]])
                    ei_j = torch.from_numpy(ei[:, cols[j]] - first[j])
                inner = copy.copy(item)
                xj = x_dev[n0:n0 + n]
                inner.x = xj.view((n,) + tuple(item.x.shape[1:])) if item.x.dim() != 2 else xj
                ticket = _Ticket(inner)
                self.__dict__["_expect"] = (ticket.x.data_ptr(), (host_abs[n0:n0 + n], rel_j, ei_j))
                n0 += n
                yield ticket
            self.__dict__["_expect"] = None
            cur = nxt


def lookahead(loader: Iterable, model, device, micro_batch: int = 64, bf16_input: Optional[bool] = None):
    """-> (loader', model') for the reference's loop: ``for batch_idx, data in enumerate(loader'): ... model'(data.to(device))``."""
    m = Lookahead(model, device, micro_batch, bf16_input)
    return m.ahead(loader), m

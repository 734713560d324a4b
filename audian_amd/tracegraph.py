"""A GUI-free stand-in for the part of audian's ``Data`` model that drives the trace graph
(``src/audian/data.py:121-236`` in /root/reference): put the traces in dependency order, add up
the pre/post-roll every level needs, open the raw loader and the derived traces, and re-align
the derived buffers whenever the visible time window moves.  Tests, the streaming demo and
integrators use it; the Qt application keeps its own ``Data``."""

from .bufferedarray import ArrayLoader

ROOT = 'data'          # name of the raw recording every chain starts from


def _widen(a, b):
    return max(a[0], b[0]), max(a[1], b[1])


class TraceGraph(object):

    def __init__(self, buffer_time=60.0, back_time=20.0):
        self.buffer_time, self.back_time = buffer_time, back_time
        self.data = None
        self.traces, self.sources = [], []
        self.tbefore = self.tafter = 0

    def add_trace(self, trace):
        self.traces.append(trace)

    def __getitem__(self, key):
        wanted = key.lower()
        return next((t for t in self.traces if t.name.lower() == wanted), None)

    def setup_traces(self):
        """Depth-first order: every trace directly after its source and before its source's
        next sibling, siblings in the order they were added -- the order data.py:121-147 builds by
        repeated insertion.  `sources[k]` is the position of trace k's source, -1 for the raw data."""
        children = {}
        for trace in self.traces:
            children.setdefault(trace.source_name, []).append(trace)
        ordered, parents = [], []

        def place(source_name, position):
            for trace in children.pop(source_name, []):
                ordered.append(trace)
                parents.append(position)
                place(trace.name, len(ordered) - 1)

        place(ROOT, -1)
        if children:
            orphans = [f'{t.name} <- {t.source_name}' for group in children.values() for t in group]
            raise ValueError('source not found for traces: ' + ', '.join(orphans))
        self.traces, self.sources = ordered, parents

    def open(self, samples, rate, unwrap=0.0, unwrap_clip=False, **kwargs):
        """What Data.open does around the loader (data.py:150-204): walk the ordered traces
        from the leaves up, let each add the margins its dependants need to its own
        (expand_times) and pass the sum on to its source; the raw loader is then opened with
        buffer_time / back_time widened by what arrives at the root, and every derived trace is
        opened on its source."""
        n = len(self.traces)
        needed = [(0, 0)]*n                 # margins the traces derived from k ask of k
        root = (0, 0)
        for k in range(n - 1, -1, -1):
            ask = self.traces[k].expand_times(*needed[k])
            parent = self.sources[k]
            if parent < 0:
                root = _widen(root, ask)
            else:
                needed[parent] = _widen(needed[parent], ask)
        self.tbefore, self.tafter = root
        self.data = ArrayLoader(samples, rate, self.buffer_time + self.tbefore + self.tafter,
                                self.back_time + self.tbefore, **kwargs)
        if unwrap > 1e-3:
            # what Data.open does right behind the loader (data.py:180; CLI -u / -U, audian.py:1485-1512)
            self.data.set_unwrap(unwrap, unwrap_clip, False, self.data.unit)
        # position 0 is the raw data from now on
        self.traces = [self.data] + self.traces
        self.sources = [None] + [p + 1 for p in self.sources]
        for trace, parent in zip(self.traces[1:], self.sources[1:]):
            trace.open(self.traces[parent])
        self.set_need_update()

    def set_need_update(self):
        """The raw data is needed if it is shown itself; the derived traces then report their own
        need, which climbs back up (BufferedData.set_need_update)."""
        raw = self.data
        raw.need_update = any(item is not None and item.isVisible() for item in raw.plot_items)
        for dest in raw.dests:
            dest.set_need_update()

    def update_times(self, t0, t1):
        """A new visible window [t0, t1] s: move the raw buffer (widened by the accumulated
        margins), then re-align every derived trace that is needed, sources first (data.py:225-231)."""
        if self.data.need_update:
            self.data.update_time(t0 - self.tbefore, t1 + self.tafter)
        for trace in self.traces[1:]:
            if trace.need_update:
                trace.align_buffer()

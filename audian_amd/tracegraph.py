"""The call order in which audian's ``Data`` model drives the trace graph
(``src/audian/data.py:121-236`` in /root/reference), without the GUI: ordering of
traces by source, accumulation of pre/post-roll times, opening, and re-alignment of
every derived trace when the visible time window moves.  Tests, the streaming
demo and integrators use it; the Qt application keeps using its own ``Data``."""

from .bufferedarray import ArrayLoader


class TraceGraph(object):

    def __init__(self, buffer_time=60.0, back_time=20.0):
        self.buffer_time = buffer_time
        self.back_time = back_time
        self.data = None
        self.traces = []
        self.sources = []
        self.tbefore = 0
        self.tafter = 0

    def add_trace(self, trace):
        self.traces.append(trace)

    def __getitem__(self, key):
        for trace in self.traces:
            if trace.name.lower() == key.lower():
                return trace
        return None

    def setup_traces(self):
        """Order traces so that every trace comes after its source (data.py:121-147)."""
        pending = list(self.traces)
        ordered, sources = [], []
        names = ['data']
        i = -1
        while i < len(ordered):
            sname = ordered[i].name if i >= 0 else 'data'
            kids = [t for t in pending if t.source_name == sname]
            pending = [t for t in pending if t.source_name != sname]
            for t in reversed(kids):
                ordered.insert(i + 1, t)
                sources.insert(i + 1, i)
            i += 1
        if pending:
            raise ValueError('source not found for traces: ' +
                             ', '.join(f'{t.name} <- {t.source_name}' for t in pending))
        self.traces, self.sources = ordered, sources
        del names

    def open(self, samples, rate, **kwargs):
        """Accumulate pre/post-roll, open the raw loader, open every derived trace
        (data.py:150-204)."""
        self.tbefore = 0
        self.tafter = 0
        tbefore = [0]*len(self.traces)
        tafter = [0]*len(self.traces)
        for k in reversed(range(len(self.traces))):
            tb, ta = self.traces[k].expand_times(tbefore[k], tafter[k])
            i = self.sources[k]
            if i < 0:
                self.tbefore = max(self.tbefore, tb)
                self.tafter = max(self.tafter, ta)
            else:
                tbefore[i] = max(tbefore[i], tb)
                tafter[i] = max(tafter[i], ta)
        tbuffer = self.buffer_time + self.tbefore + self.tafter
        tback = self.back_time + self.tbefore
        self.data = ArrayLoader(samples, rate, tbuffer, tback, **kwargs)
        self.traces.insert(0, self.data)
        self.sources = [None] + [i + 1 for i in self.sources]
        for trace, source in zip(self.traces[1:], self.sources[1:]):
            trace.open(self.traces[source])
        self.set_need_update()

    def set_need_update(self):
        self.data.need_update = False
        for pi in self.data.plot_items:
            if pi is not None and pi.isVisible():
                self.data.need_update = True
                break
        for d in self.data.dests:
            d.set_need_update()

    def update_times(self, t0, t1):
        """Move the raw buffer, then re-align every derived trace (data.py:225-231)."""
        if self.data.need_update:
            self.data.update_time(t0 - self.tbefore, t1 + self.tafter)
        for trace in self.traces[1:]:
            if trace.need_update:
                trace.align_buffer()

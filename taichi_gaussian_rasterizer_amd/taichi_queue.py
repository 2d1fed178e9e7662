"""TaichiQueue compatibility shim.

The reference serialises every Taichi launch on one worker thread (taichi_queue.py:39-90) and
callers bracket their programs with TaichiQueue.init(...) / taichi_queue(...) / TaichiQueue.stop()
(examples/fit_image_gaussians.py:249, tests/test_rasterizer.py:63).  There is no Taichi here:
kernels are launched on the caller's HIP stream and the C-ABI is re-entrant, so these calls are
accepted and run the function inline.
"""
from __future__ import annotations

from concurrent.futures import Future


class TaichiQueueContext:
    def __init__(self, *args, **kwargs):
        self.args, self.kwargs = args, kwargs

    def __enter__(self):
        TaichiQueue.init(*self.args, **self.kwargs)

    def __exit__(self, exc_type, exc_value, traceback):
        TaichiQueue.stop()


def taichi_queue(*args, **kwargs):
    return TaichiQueueContext(*args, **kwargs)


class TaichiQueue:
    executor = None

    @classmethod
    def init(cls, *args, threaded=False, **kwargs) -> None:
        cls.executor = "inline"

    @staticmethod
    def thread_id():
        return None

    @classmethod
    def queue(cls):
        return cls.executor

    @staticmethod
    def run_async(func, *args, **kwargs) -> Future:
        future = Future()
        args = [a.result() if isinstance(a, Future) else a for a in args]
        future.set_result(func(*args, **kwargs))
        return future

    @staticmethod
    def run_sync(func, *args, **kwargs):
        return TaichiQueue.run_async(func, *args, **kwargs).result()

    @classmethod
    def stop(cls) -> None:
        cls.executor = None


def queued(kernel):
    def f(*args, **kwargs):
        return kernel(*args, **kwargs)
    return f

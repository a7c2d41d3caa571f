"""`SynchronizedBatchNorm2d` as the reference's training actually runs it.

The reference vendors the DataParallel-callback implementation (models/networks/sync_batchnorm/batchnorm.py).  It only
synchronises inside `DataParallelWithCallback`; under Lightning DDP — one process per GPU, which is also this
package's execution model — `_is_parallel` stays False and `forward` is a plain `F.batch_norm` on the local batch
(batchnorm.py:63-68): per-rank statistics, running statistics updated, `num_batches_tracked` NOT incremented (the
functional form never touches it).  That is what this class does; no cross-rank exchange is invented.
"""
from .layers import HipBatchNorm2d


class SynchronizedBatchNorm2d(HipBatchNorm2d):
    def __init__(self, num_features, eps=1e-5, momentum=0.1, affine=True):
        super().__init__(num_features, eps=eps, momentum=momentum, affine=affine, count_batches=False)

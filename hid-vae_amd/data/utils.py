"""cycle / batch_to / next_batch helpers of the reference (data/utils.py:3-37) for the (Tagged)SeqBatch records."""
from .schemas import SeqBatch, TaggedSeqBatch


def cycle(dataloader):
    while True:
        for data in dataloader:
            yield data


def batch_to(batch, device):
    if isinstance(batch, (SeqBatch, TaggedSeqBatch)):
        return type(batch)(*[t.to(device) if hasattr(t, "to") else t for t in batch])
    return batch.to(device)


def next_batch(dataloader, device):
    return batch_to(next(dataloader), device)

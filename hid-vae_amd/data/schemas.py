"""Batch / result records of the tokenizer path, field-for-field what the reference's callers unpack
(reference data/schemas.py:7-24 SeqBatch/TaggedSeqBatch, :27-45 tokenized batches, :57-69 HRqVaeComputedLosses,
:74-97 HRqVaeOutput).  Built with collections.namedtuple so positional construction keeps working."""
from collections import namedtuple

FUT_SUFFIX = "_fut"

_SEQ = ("user_ids", "ids", "ids_fut", "x", "x_fut", "seq_mask")
_TOK = ("user_ids", "sem_ids", "sem_ids_fut", "seq_mask", "token_type_ids", "token_type_ids_fut")
_TAGS = ("tags_emb", "tags_indices")

SeqBatch = namedtuple("SeqBatch", _SEQ)
TaggedSeqBatch = namedtuple("TaggedSeqBatch", _SEQ + _TAGS)
TokenizedSeqBatch = namedtuple("TokenizedSeqBatch", _TOK)
TaggedTokenizedSeqBatch = namedtuple("TaggedTokenizedSeqBatch", _TOK + _TAGS)

_LOSS_FIELDS = ("loss", "reconstruction_loss", "rqvae_loss", "tag_align_loss", "tag_pred_loss", "tag_pred_accuracy",
                "embs_norm", "p_unique_ids", "tag_align_loss_by_layer", "tag_pred_loss_by_layer",
                "tag_pred_accuracy_by_layer", "sem_id_uniqueness_loss")
HRqVaeComputedLosses = namedtuple("HRqVaeComputedLosses", _LOSS_FIELDS, defaults=(None, None, None, None))


class HRqVaeOutput:
    """What get_semantic_ids returns: embeddings [B,D,L], residuals [B,D,L], sem_ids [B,L] + the level losses."""

    __slots__ = ("embeddings", "residuals", "sem_ids", "quantize_loss", "tag_align_loss", "tag_pred_loss",
                 "tag_pred_accuracy", "tag_align_loss_by_layer", "tag_pred_loss_by_layer", "tag_pred_accuracy_by_layer")

    def __init__(self, embeddings, residuals, sem_ids, quantize_loss, tag_align_loss, tag_pred_loss, tag_pred_accuracy,
                 tag_align_loss_by_layer=None, tag_pred_loss_by_layer=None, tag_pred_accuracy_by_layer=None):
        for name, value in zip(self.__slots__, (embeddings, residuals, sem_ids, quantize_loss, tag_align_loss, tag_pred_loss,
                                                tag_pred_accuracy, tag_align_loss_by_layer, tag_pred_loss_by_layer,
                                                tag_pred_accuracy_by_layer)):
            setattr(self, name, value)

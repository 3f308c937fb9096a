from .batch_data import adjust_batch_size, get_batch_loader
from .batch_unit import PairFeats, PairwiseBatch, PointwiseBatch, PointwiseSepFeatBatch, SeqFeats, TripleFeats
from .collators import BaseCollator, PairwiseCollator, PointwiseCollator
from .sequence import SequenceBuilder, get_interacted_seqs, get_recent_seqs

__all__ = ["adjust_batch_size", "get_batch_loader", "PairFeats", "PairwiseBatch", "PointwiseBatch",
           "PointwiseSepFeatBatch", "SeqFeats", "TripleFeats", "BaseCollator", "PairwiseCollator",
           "PointwiseCollator", "SequenceBuilder", "get_interacted_seqs", "get_recent_seqs"]

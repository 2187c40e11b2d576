"""ppo_amd — MI355X-native experience path of flawedmatrix/ppo (rollout buffer -> GAE/returns ->
advantage standardisation -> shuffled minibatch gather) behind the reference's own interface.

Compute lives in libppo_xp.so (hand-written HIP for gfx950, C ABI in include/ppo_xp.h).  The
classes here mirror ppo/src/common/buffer.rs and ppo/src/model/data.rs.  There is no CPU path.
"""
from ._lib import GAE_ASSOC, GAE_NO_PACK, GAE_SEQUENTIAL, PpoXpError, PpoXpPanic, gae_chunk
from .buffer import ExperienceBuffer, device_tensor, pinned_empty
from .data import ExperienceBatch, ExperienceBatcher, ExperienceBatcherIterator, HostBatch
from .runner import RewardNormalizer

__all__ = ["ExperienceBuffer", "ExperienceBatch", "ExperienceBatcher", "ExperienceBatcherIterator", "HostBatch", "RewardNormalizer", "PpoXpError",
           "PpoXpPanic", "GAE_SEQUENTIAL", "GAE_ASSOC", "GAE_NO_PACK", "gae_chunk", "device_tensor", "pinned_empty"]

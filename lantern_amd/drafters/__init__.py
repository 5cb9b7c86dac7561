"""Host mirror of the reference's models/drafters package for the verify/accept path:
tree shapes (choices), KV cache bookkeeping (kv_cache) and the drafter-side tree ops (tree)."""

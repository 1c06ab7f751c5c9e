"""GPT-2 embeddings -- mirror of the reference's flash_attn/modules/embedding.py:11-39."""
import torch
import torch.nn as nn


class GPT2Embeddings(nn.Module):
    """word (+ learned absolute position) embeddings; max_position_embeddings <= 0 disables
    the position table."""

    def __init__(self, embed_dim, vocab_size, max_position_embeddings, padding_idx=None,
                 device=None, dtype=None):
        factory_kwargs = {'device': device, 'dtype': dtype}
        super().__init__()
        self.word_embeddings = nn.Embedding(vocab_size, embed_dim, padding_idx=padding_idx,
                                            **factory_kwargs)
        self.max_position_embeddings = max_position_embeddings
        if max_position_embeddings > 0:
            self.position_embeddings = nn.Embedding(max_position_embeddings, embed_dim,
                                                    **factory_kwargs)

    def forward(self, input_ids, position_ids=None):
        emb = self.word_embeddings(input_ids)
        if self.max_position_embeddings > 0:
            if position_ids is None:
                position_ids = torch.arange(input_ids.shape[1], dtype=torch.long,
                                            device=input_ids.device)
            emb = emb + self.position_embeddings(position_ids)
        return emb

"""Stand-in for the reference's Rust wheel so that `interaction_consumed`
(libreco/data/consumed.py:13-17) gets the canonical semantics of rust/src/utils.rs:8-35:
append in order, then drop only CONSECUTIVE repeats (pinned by tests/test_consumed.py:12-25)."""


def build_consumed_unique(user_indices, item_indices):
    uc, ic = {}, {}
    for u, i in zip(user_indices, item_indices):
        uc.setdefault(u, []).append(i)
        ic.setdefault(i, []).append(u)

    def dedup(v):
        out = [v[0]]
        for x in v[1:]:
            if x != out[-1]:
                out.append(x)
        return out

    return {k: dedup(v) for k, v in uc.items()}, {k: dedup(v) for k, v in ic.items()}

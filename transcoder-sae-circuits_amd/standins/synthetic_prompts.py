"""IOI-shaped prompt sets for tests and benchmarks (stand-in: the reference takes them from
``data.ioi_dataset.gen_templated_prompts`` (src/data/ioi_dataset.py:465-484), whose template and name tables stay
in the reference).  Same record layout (``text`` / ``correct`` / ``counter``, ``data/eval_dataset.py``), every
prompt of a set has the same number of words, so a whitespace tokenizer yields equal lengths."""
from __future__ import annotations

import random
from typing import Dict, List

_NAMES = ["Ada", "Ben", "Cleo", "Dov", "Eve", "Finn", "Gus", "Hana", "Ivo", "Jun", "Kai", "Lena", "Milo", "Nora",
          "Otto", "Pia", "Quin", "Rosa", "Sven", "Tess", "Uma", "Vik", "Wren", "Xan", "Yara", "Zed"]
_PLACES = ["harbour", "library", "market", "station", "orchard", "workshop", "museum", "bakery"]
_OBJECTS = ["lantern", "notebook", "kettle", "compass", "scarf", "whistle", "marble", "ribbon"]


def make_ioi_prompts(n: int, seed: int = 0) -> List[Dict[str, str]]:
    rng = random.Random(seed)
    out = []
    for _ in range(n):
        a, b = rng.sample(_NAMES, 2)
        place, obj = rng.choice(_PLACES), rng.choice(_OBJECTS)
        giver, receiver = (a, b) if rng.random() < 0.5 else (b, a)
        text = f"Later {a} and {b} spent the evening at the {place} and {giver} handed a {obj} to"
        out.append({"text": text, "correct": " " + receiver, "counter": " " + giver})
    return out

"""Parity ledger: the GPU parity tests record what they measured (not just pass / fail) into
``gpurun_out/parity_ledger.json`` on the box that ran them; the builder copies it to ``profiles/rNN_parity.json``."""
import json
import os

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PATH = os.path.join(REPO, "gpurun_out", "parity_ledger.json")


def record(section: str, key: str, entry: dict) -> None:
    try:
        os.makedirs(os.path.dirname(PATH), exist_ok=True)
        data = {}
        if os.path.exists(PATH):
            with open(PATH) as f:
                data = json.load(f)
        data.setdefault(section, {})[key] = entry
        with open(PATH, "w") as f:
            json.dump(data, f, indent=1, sort_keys=True)
    except OSError:
        pass  # read-only checkout: the assertions still run

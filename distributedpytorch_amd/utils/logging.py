"""Rank-0 structured logging.  The reference prints to stdout and pushes one
figure per epoch to a hosted service with a hard-coded key (`/root/reference/train_pascal.py:41`,
`:276`); nothing here talks to any external service.  Records go to ``<run>/log.jsonl`` and a
one-line stdout summary; ``generate_param_report`` (`:169`) lives in ``data.helpers``."""
from __future__ import annotations

import json
import os
import sys
import time
from typing import Any, Dict, Optional

from ..parallel import dist_env


class JsonlLogger:
    def __init__(self, path: Optional[str], echo: bool = True):
        self.path = path if dist_env.is_master() else None
        self.echo = echo and dist_env.is_master()
        self._t0 = time.time()
        if self.path:
            os.makedirs(os.path.dirname(os.path.abspath(self.path)), exist_ok=True)
            self._fh = open(self.path, "a", buffering=1)
        else:
            self._fh = None

    def log(self, kind: str, **fields: Any) -> None:
        rec: Dict[str, Any] = {"t": round(time.time() - self._t0, 3), "kind": kind}
        rec.update(fields)
        if self._fh:
            self._fh.write(json.dumps(rec, default=float) + "\n")
        if self.echo:
            body = " ".join(f"{k}={_fmt(v)}" for k, v in fields.items())
            print(f"[{rec['t']:9.2f}s] {kind}: {body}", file=sys.stdout, flush=True)

    def close(self) -> None:
        if self._fh:
            self._fh.close()
            self._fh = None


def _fmt(v: Any) -> str:
    if isinstance(v, float):
        return f"{v:.6g}"
    return str(v)

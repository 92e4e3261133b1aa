"""Logger, execution-time decorator and loss CSV writer (deblur_denoise/utils/logging_utils.py).
Observability only; nothing here touches the GPU."""
from __future__ import annotations

import logging
import os
import time
from functools import wraps
from pathlib import Path
from typing import Dict, List, Optional

CO_FILE_LOGLEVEL = logging.DEBUG if os.environ.get("CO_FILE_LOGLEVEL") == "DEBUG" else logging.INFO
CO_CONSOLE_LOGLEVEL = logging.DEBUG if os.environ.get("CO_CONSOLE_LOGLEVEL") == "DEBUG" else logging.WARNING

logger = logging.getLogger("convex_optimization_amd")
if not logger.handlers:
    logger.setLevel(logging.DEBUG)
    _h = logging.StreamHandler()
    _h.setLevel(CO_CONSOLE_LOGLEVEL)
    _h.setFormatter(logging.Formatter("%(asctime)s - %(levelname)s - %(message)s"))
    logger.addHandler(_h)


def log_execution_time(log: logging.Logger):
    """Logs wall-clock seconds of the wrapped call (the reference logs process CPU time, lu:63-86)."""
    def deco(fn):
        @wraps(fn)
        def wrapper(*a, **k):
            t0 = time.perf_counter()
            try:
                return fn(*a, **k)
            finally:
                log.info(f"{fn.__name__} took {time.perf_counter() - t0:.4f} s")
        return wrapper
    return deco


def save_loss_data(loss_list: List[float], algorithm_name: str, loss_function_name: str,
                   parameters: Optional[Dict] = None, start_time: Optional[float] = None,
                   results_dir: Optional[str] = None) -> str:
    """Per-iteration metrics as results/{alg}_{loss}_{int(start)}.csv with the reference's columns
    (lu:88-138): loss, algorithm, loss_function, time_taken, start_time, <scalar parameters>, and the
    per-iteration psnr / ssim / mse / l1_loss / l2_loss lists."""
    import pandas as pd
    start_time = start_time if start_time is not None else time.time()
    out_dir = Path(results_dir) if results_dir else Path.cwd() / "results"
    out_dir.mkdir(parents=True, exist_ok=True)
    df = pd.DataFrame({"loss": loss_list})
    df["algorithm"] = algorithm_name
    df["loss_function"] = loss_function_name
    df["time_taken"] = time.time() - start_time
    df["start_time"] = start_time
    for key, val in (parameters or {}).items():
        if isinstance(val, (list, tuple)):
            if len(val) == len(loss_list):
                df[key] = list(val)
        else:
            df[key] = val
    path = out_dir / f"{algorithm_name}_{loss_function_name}_{int(start_time)}.csv"
    if path.exists():
        df.to_csv(path, mode="a", header=False, index=False)     # the reference appends to an existing file (lu:128-131)
        logger.info(f"Loss data appended to existing file {path}")
    else:
        df.to_csv(path, index=False)
        logger.info(f"Loss data saved to new file {path}")
    return str(path)

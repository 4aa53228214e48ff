"""
Log level of the package (mirror of the reference's log.py:9-45, on the standard ``logging`` module: loguru is not a
dependency here).  ``DEBUG`` has one effect on the hot path, like in the reference: ``create_sequential_module`` puts a
``DetectAnomaly`` layer behind every module (model_factory/utils.py:85-87), which checks every tensor of the data dict
for NaN / Inf -- one host synchronisation per tensor, so it is for debugging only.
"""
import logging
import sys
from typing import Optional

LOG_LEVEL: Optional[str] = None
logger = logging.getLogger("matten_amd")


def set_logger(level: str = "INFO", filename: Optional[str] = None, stderr: bool = True) -> None:
    """level: DEBUG, INFO, WARNING, ERROR or CRITICAL; filename: also log to this file (the reference writes matten.log)"""
    global LOG_LEVEL
    level = level.upper()
    if level not in ("DEBUG", "INFO", "WARNING", "ERROR", "CRITICAL"):
        raise ValueError(f"unknown log level {level}")
    LOG_LEVEL = level
    logger.setLevel(level)
    for h in list(logger.handlers):
        logger.removeHandler(h)
    if stderr:
        logger.addHandler(logging.StreamHandler(sys.stderr))
    if filename:
        logger.addHandler(logging.FileHandler(filename))


def get_log_level() -> Optional[str]:
    return LOG_LEVEL

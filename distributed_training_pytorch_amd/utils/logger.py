"""Rank-aware file+stream logger.

Parity target: reference ``utils/logger.py:5-33`` — ``Logger(log_name, file)`` with
``.log(message, log_type)`` where ``log_type`` in {"warning", "error"} selects that level and
anything else is INFO; record format ``'%(asctime)s - %(name)s - %(levelname)s - %(message)s'``
with the reference's three-space date format.

Differences (fixed on purpose):
  * the parent directory of ``file`` is created (the reference crashes on a fresh checkout);
  * only rank 0 truncates/owns the shared log file; other ranks either stay silent
    (``all_ranks=False``, the default) or write to ``<file>.rank<k>``;
  * re-creating a Logger with the same name does not stack duplicate handlers.
"""
import logging
import os

_FMT = "%(asctime)s - %(name)s - %(levelname)s - %(message)s"
_DATEFMT = "%Y-%m-%d   %H:%M:%S"


def _env_rank():
    try:
        return int(os.environ.get("RANK", "0"))
    except ValueError:
        return 0


class Logger(object):
    def __init__(self, log_name, file=None, all_ranks=False, rank=None, stream=True):
        self.rank = _env_rank() if rank is None else int(rank)
        self.all_ranks = bool(all_ranks)
        self.enabled = self.all_ranks or self.rank == 0
        name = log_name if self.rank == 0 else "%s[r%d]" % (log_name, self.rank)
        self.logger = logging.getLogger(name)
        self.logger.setLevel(logging.INFO)
        self.logger.propagate = False
        for h in list(self.logger.handlers):  # idempotent re-construction
            self.logger.removeHandler(h)
            try:
                h.close()
            except Exception:
                pass
        self.file = None
        if not self.enabled:
            self.logger.addHandler(logging.NullHandler())
            return
        form = logging.Formatter(fmt=_FMT, datefmt=_DATEFMT)
        if stream:
            sh = logging.StreamHandler()
            sh.setFormatter(form)
            self.logger.addHandler(sh)
        if file is not None:
            path = file if self.rank == 0 else "%s.rank%d" % (file, self.rank)
            parent = os.path.dirname(os.path.abspath(path))
            os.makedirs(parent, exist_ok=True)
            # mode="w" truncates: same observable effect as the reference's delete-then-create
            fh = logging.FileHandler(path, mode="w")
            fh.setFormatter(form)
            self.logger.addHandler(fh)
            self.file = path

    def log(self, message, log_type="info"):
        if not self.enabled:
            return
        if log_type == "warning":
            self.logger.warning(message)
        elif log_type == "error":
            self.logger.error(message)
        else:
            self.logger.info(message)

    def close(self):
        for h in list(self.logger.handlers):
            try:
                h.flush()
                h.close()
            except Exception:
                pass
            self.logger.removeHandler(h)

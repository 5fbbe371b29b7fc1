"""CSVLogger: the on-disk wire format of a benchmark run, one file per (MDP, agent, seed)
(reference colosseum/utils/acme/csv_logger.py:20-135 as run_experiment_instance uses it, experiment_instances.py:205-210:
`<directory>/logs/<label>/<file_name>.csv`, header = the sorted keys of the first row, one line per logging step, values
printed as `str(np.array(v))`, csv module line endings).  `colosseum.analysis` reads these files; the batched runner
writes the same text from its column store (`vector_tracker.BatchLog.csv_text`)."""
import csv
import os

import numpy as np


class CSVLogger:
    def __init__(self, directory: str, label: str = "", file_name: str = "logs"):
        self._directory = os.path.join(directory, "logs", label) if label else os.path.join(directory, "logs")
        os.makedirs(self._directory, exist_ok=True)
        self.file_path = os.path.join(self._directory, f"{file_name}.csv")
        self._file = open(self.file_path, "w", newline="")
        self._writer = None

    def reset(self):
        """MDPLoop.run calls it before the first row (the reference's CSVLogger.reset re-opens the file)."""
        self._file.seek(0)
        self._file.truncate()
        self._writer = None

    def write(self, data):
        if self._writer is None:
            self._writer = csv.DictWriter(self._file, fieldnames=sorted(data.keys()), extrasaction="ignore")
            self._writer.writeheader()
        self._writer.writerow({k: np.array(v) for k, v in data.items()})

    def flush(self):
        self._file.flush()

    def close(self):
        self._file.flush()
        self._file.close()

"""``from mypath import Path`` — dataset root lookup the reference expects
(`/root/reference/pascal.py:13`, `:33`).  Roots come from the environment, never from source."""
import os


class Path(object):
    @staticmethod
    def db_root_dir(database: str = "pascal") -> str:
        env = {"pascal": "VOC_ROOT", "sbd": "SBD_ROOT"}.get(database)
        if env is None:
            raise NotImplementedError(f"Database {database} not available.")
        return os.environ.get(env, os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", database))

    @staticmethod
    def models_dir() -> str:
        return os.environ.get("DPAMD_MODELS", os.path.join(os.path.dirname(os.path.abspath(__file__)), "models"))

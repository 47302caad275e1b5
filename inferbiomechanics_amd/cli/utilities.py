"""git helpers used for run metadata (role of src/cli/utilities.py:5-20)."""
import subprocess


def _git(*a) -> str:
    try:
        return subprocess.run(['git', *a], capture_output=True, text=True, timeout=10).stdout.strip()
    except Exception:
        return ''


def get_git_hash() -> str:
    return _git('rev-parse', 'HEAD')


def has_uncommitted_changes() -> bool:
    return _git('status', '--porcelain') != ''

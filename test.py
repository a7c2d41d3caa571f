"""Reference entry point `python test.py --model ... --checkpoint ...` (test.py:9-10): see shineon_virtual_tryon_amd/cli.py."""
import shineon_virtual_tryon_amd  # noqa: F401  (registers the hyphenated package directory)
from shineon_virtual_tryon_amd.cli import run

if __name__ == "__main__":
    run(train=False)

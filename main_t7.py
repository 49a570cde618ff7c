"""Alias of main.py under the reference's PyTorch entry-point name (/root/reference/main_t7.py)."""
import sys

from main import run

if __name__ == '__main__':
    run(sys.argv[1:])

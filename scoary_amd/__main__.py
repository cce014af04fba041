"""``python -m scoary_amd -g genes.csv -t traits.csv [--no_pairwise] [--permute N] ...``"""
from .methods import main

if __name__ == "__main__":
    main()

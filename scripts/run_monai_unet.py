"""Same entry as the reference's scripts/run_monai_unet.py:1-4."""
from segmantic_amd.commands.monai_unet_cli import main

if __name__ == "__main__":
    main()

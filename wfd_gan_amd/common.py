"""Constants shared by the host scripts (same names/values as the reference's src/common.py:1-19)."""
from os.path import abspath, dirname, join, pardir

BASE_DIR = abspath(join(dirname(__file__), pardir))
confdir = join(BASE_DIR, "conf.ini")
outputdir = join(BASE_DIR, "dump")
dModelDir = join(BASE_DIR, "f_model")
LOG_FORMAT = "%(asctime)s %(name)-12s %(levelname)-8s %(message)s"

lambda_gp = 10          # WGAN-GP weight (src/common.py:10)

# generator checkpoint name template (src/common.py:19); consumers parse seqlen / cls / latentdim out of it
GENERATOR_NAME_REG = r"generator_seqlen(?P<seqlen>\d+)_cls(?P<cls>\d+)_latentdim(?P<latentdim>\d+).ckpt"

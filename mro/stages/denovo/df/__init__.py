"""Stage code of ASSEMBLER_DF as Martian loads it (`src py "stages/denovo/df"`, mro/_assembler_df_gpu.mro): the three
entry points of the reference's mro/stages/denovo/df/__init__.py (split :8-12, main :81-173, join :14-15), implemented
in supernova_amd.df_stage (the graph is built on the MI355X when `mspedges` is null)."""
from supernova_amd.df_stage import join, main, split  # noqa: F401

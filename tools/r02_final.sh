# usage: bash tools/r02_final.sh <tag>  -- GPU box: everything the round's profile set holds, from one build:
# GPU suite + default bench + rocprofv3 stats + serial timeline + PMC passes + 1S+1T line (r02_full.sh), the twitter / uk
# shapes and the other sample types (r02_shapes.sh), the pipeline stages alone (r02_stages.sh)
tag=$1
bash tools/r02_full.sh $tag pytest || exit 1
bash tools/r02_shapes.sh $tag || exit 1
bash tools/r02_stages.sh $tag

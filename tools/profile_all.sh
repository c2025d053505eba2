# usage: bash tools/profile_all.sh <tag>  -- GPU box: everything the round's profile set holds, from one build:
# GPU suite + default bench + rocprofv3 stats + serial timeline + PMC passes + 1S+1T line (profile_full.sh), the twitter / uk
# shapes and the other sample types (profile_shapes.sh), the pipeline stages alone (profile_stages.sh)
tag=$1
bash tools/profile_full.sh $tag pytest || exit 1
bash tools/profile_shapes.sh $tag || exit 1
bash tools/profile_stages.sh $tag

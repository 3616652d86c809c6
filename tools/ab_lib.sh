L=sr-caco-2_amd/lib
for i in 1 2; do
python bench.py 2>&1 | tail -1 | cut -c60-110
python tools/mb_tn_roles.py 2>&1 | grep "DBG=0"
cp $L/libsrhip.so $L/new.so; cp $L/libsrhip_old.so $L/libsrhip.so
echo OLD; python bench.py 2>&1 | tail -1 | cut -c60-110
python tools/mb_tn_roles.py 2>&1 | grep "DBG=0"
cp $L/new.so $L/libsrhip.so; echo NEW
done

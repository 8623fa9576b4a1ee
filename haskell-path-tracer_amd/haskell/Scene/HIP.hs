{-# LANGUAGE ForeignFunctionInterface #-}
{-# LANGUAGE BangPatterns #-}
{-# LANGUAGE RecordWildCards #-}

-- | Drop-in replacement for the Accelerate-compiled render function of
-- haskell-path-tracer: binds libptmi (include/ptmi.h) and produces the very
-- 'CompiledFunction' that @compileFor@ builds at app/Main.hs:188-191, plus the
-- two array programs that create / replace the RNG planes (app/Main.hs:155,
-- :231, :306).
--
-- SOURCE ONLY: the build container has no GHC, so this module has not been
-- compiled here.  It uses nothing beyond @base@, @vector@ and the packages the
-- application already depends on (accelerate, accelerate-io-vector, linear).
--
-- To switch the application over, in app/Main.hs:
--
-- > import qualified Scene.HIP as HIP
-- > ...
-- > hip <- HIP.initialise 0                    -- once, in main, before :154
-- > let compute' = HIP.compileFor hip arguments -- replaces `compileFor arguments` (:154)
-- > seeds <- HIP.initialOutput hip 0x5EED1234   -- replaces `run <$> initialOutput` (:155, :306)
-- > reseeded <- HIP.reseed hip seed0 acc        -- replaces `run <$> (reseed . A.use $ acc)` (:231)
--
-- Everything else (threads, MVar, SDL/GL presentation) stays as it is: the
-- closure still has type Camera -> (Int, RenderResult) -> (Int, RenderResult).
module Scene.HIP
  ( Handle
  , initialise
  , compileFor
  , initialOutput
  , reseed
  , PtmiError(..)
  ) where

import           Control.Exception
import           Control.Monad                  ( when )
import qualified Data.Array.Accelerate         as A
import           Data.Array.Accelerate.IO.Data.Vector.Storable
                                                ( fromVectors, toVectors )
import qualified Data.Vector.Storable          as V
import qualified Data.Vector.Storable.Mutable  as VM
import           Data.Int
import           Data.Word
import           Foreign
import           Foreign.C.String
import           Foreign.C.Types
import           GHC.Float                      ( castWord32ToFloat )
import           Linear                         ( V3(..) )
import           System.IO.Unsafe               ( unsafePerformIO )

import           Scene.Objects
import           Scene.Trace                    ( Algorithm(..) )
import           Scene.World                    ( mainScene' )   -- see note [scene as data]
import           Util                           ( screenWidth, screenHeight )

-- Note [scene as data]
-- Scene.World.mainScene is a list of Exp constants baked into the Accelerate
-- kernel (src/Scene/World.hs:15-77).  libptmi takes the scene at run time, so
-- World.hs additionally exports the plain Haskell values it already contains:
--
-- > mainScene' :: ([Sphere], [Plane])
-- > mainScene' = (spheres', planes')   -- the two where-bound lists, lifted to top level

data PtmiCtx
newtype Handle = Handle (ForeignPtr PtmiCtx)

data PtmiError = PtmiError Int String deriving Show
instance Exception PtmiError

-- ptmi_render1 and friends run for milliseconds to seconds: `safe`, so that the
-- graphics and input threads keep running (app/Main.hs:178-180).
foreign import ccall safe "ptmi.h ptmi_create"      c_create      :: Ptr (Ptr PtmiCtx) -> CInt -> IO CInt
foreign import ccall safe "ptmi.h &ptmi_destroy"    p_destroy     :: FunPtr (Ptr PtmiCtx -> IO ())
foreign import ccall safe "ptmi.h ptmi_last_error"  c_last_error  :: Ptr PtmiCtx -> IO CString
foreign import ccall safe "ptmi.h ptmi_set_scene"   c_set_scene   :: Ptr PtmiCtx -> Ptr Float -> CInt -> Ptr Float -> CInt -> IO CInt
foreign import ccall safe "ptmi.h ptmi_resize"      c_resize      :: Ptr PtmiCtx -> CInt -> CInt -> IO CInt
foreign import ccall safe "ptmi.h ptmi_init_output" c_init_output :: Ptr PtmiCtx -> Word64 -> IO CInt
foreign import ccall safe "ptmi.h ptmi_reseed"      c_reseed      :: Ptr PtmiCtx -> Word64 -> IO CInt
foreign import ccall safe "ptmi.h ptmi_upload_state"   c_upload   :: Ptr PtmiCtx -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> IO CInt
foreign import ccall safe "ptmi.h ptmi_download_state" c_download :: Ptr PtmiCtx -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> IO CInt
foreign import ccall safe "ptmi.h ptmi_render1"     c_render1
  :: Ptr PtmiCtx -> Ptr CamRec -> CInt -> CInt -> CInt -> CInt -> Ptr Int64 -> Ptr Int64
  -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> Ptr Word32
  -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> Ptr Word32
  -> IO CInt

-- | struct ptmi_camera { float position[3]; float rotation[3]; int64_t fov; }  (32 bytes)
data CamRec
pokeCamera :: Ptr CamRec -> Camera -> IO ()
pokeCamera p Camera{..} = do
  let V3 px py pz = _cameraPosition
      V3 rx ry rz = _cameraRotation
  pokeArray (castPtr p) [px, py, pz, rx, ry, rz :: Float]
  pokeByteOff p 24 (fromIntegral _cameraFov :: Int64)

check :: Ptr PtmiCtx -> CInt -> IO ()
check ctx rc = when (rc /= 0) $ do
  msg <- c_last_error ctx >>= peekCString
  throwIO (PtmiError (fromIntegral rc) msg)       -- the reference throws from inside runN as well

-- | Flat records of include/ptmi.h: sphere = 10 words, plane = 12 words.
sphereWords :: Sphere -> [Float]
sphereWords (Sphere (V3 x y z) r (Material (V3 cr cg cb) i b)) = [x, y, z, r, cr, cg, cb, i, tagBits b, param b]
planeWords :: Plane -> [Float]
planeWords (Plane (V3 x y z) (V3 nx ny nz) (Material (V3 cr cg cb) i b)) = [x, y, z, nx, ny, nz, cr, cg, cb, i, tagBits b, param b]
tagBits, param :: Brdf -> Float
tagBits (Matte _)  = castWord32ToFloat 0        -- PTMI_MATTE  (int32 tag stored in a float slot, bit pattern)
tagBits (Glossy _) = castWord32ToFloat 1        -- PTMI_GLOSSY
param (Matte p)  = p
param (Glossy p) = p

-- | Create the context, upload mainScene, size it to screenWidth x screenHeight (src/Util.hs:186-188).
initialise :: Int -> IO Handle
initialise device = alloca $ \pp -> do
  rc <- c_create pp (fromIntegral device)
  when (rc /= 0) $ c_last_error nullPtr >>= peekCString >>= throwIO . PtmiError (fromIntegral rc)
  ctx <- peek pp
  fp  <- newForeignPtr p_destroy ctx
  let (spheres, planes) = mainScene'
  -- the records travel as raw 32-bit words (the BRDF tag is an int32 bit pattern in a float slot): no conversion
  withArray (concatMap sphereWords spheres) $ \ps ->
    withArray (concatMap planeWords planes) $ \pp' ->
      c_set_scene ctx ps (fromIntegral $ length spheres) pp' (fromIntegral $ length planes) >>= check ctx
  c_resize ctx (fromIntegral screenWidth) (fromIntegral screenHeight) >>= check ctx
  return (Handle fp)

type CompiledFunction = Camera -> (Int, RenderResult) -> (Int, RenderResult)

nPixels :: Int
nPixels = fromIntegral screenWidth * fromIntegral screenHeight

-- | The seven planes of a RenderResult, in libptmi's order (r, g, b, a, b, c, counter).
-- Accelerate's representation of Matrix (V3 Float, SFC32) as nested pairs of vectors is the one
-- visible at app/Main.hs:350.
planesOf :: RenderResult -> (V.Vector Float, V.Vector Float, V.Vector Float, V.Vector Word32, V.Vector Word32, V.Vector Word32, V.Vector Word32)
planesOf acc = let (((), ((((), r), g), b)), (((((), sa), sb), sc), sd)) = toVectors acc in (r, g, b, sa, sb, sc, sd)

fromPlanes :: (V.Vector Float, V.Vector Float, V.Vector Float, V.Vector Word32, V.Vector Word32, V.Vector Word32, V.Vector Word32) -> RenderResult
fromPlanes (r, g, b, sa, sb, sc, sd) =
  fromVectors (A.Z A.:. fromIntegral screenHeight A.:. fromIntegral screenWidth)
              (((), ((((), r), g), b)), (((((), sa), sb), sc), sd))

-- | @compileFor@ (app/Main.hs:188-191) on libptmi: one call = one sample, exactly `runN (render config) screenPixels`.
-- The closure is pure from the caller's point of view, like `dewit`.
compileFor :: Handle -> Algorithm -> CompiledFunction
compileFor (Handle fp) !config = \(!c) (!iterations, !acc) ->
  (iterations + 1, unsafePerformIO (render1 c acc))
 where
  algorithm = case config of { Streams -> 0; Inline -> 1 }   -- PTMI_STREAMS / PTMI_INLINE
  render1 cam acc = withForeignPtr fp $ \ctx -> allocaBytes 32 $ \pc -> do
    pokeCamera pc cam
    let (r, g, b, sa, sb, sc, sd) = planesOf acc
    [r', g', b']        <- mapM (const $ VM.new nPixels) [1 .. 3 :: Int]
    [sa', sb', sc', sd'] <- mapM (const $ VM.new nPixels) [1 .. 4 :: Int]
    rc <- V.unsafeWith r $ \pr -> V.unsafeWith g $ \pg -> V.unsafeWith b $ \pb ->
          V.unsafeWith sa $ \pa -> V.unsafeWith sb $ \pbb -> V.unsafeWith sc $ \pcc -> V.unsafeWith sd $ \pd ->
          VM.unsafeWith r' $ \qr -> VM.unsafeWith g' $ \qg -> VM.unsafeWith b' $ \qb ->
          VM.unsafeWith sa' $ \qa -> VM.unsafeWith sb' $ \qbb -> VM.unsafeWith sc' $ \qcc -> VM.unsafeWith sd' $ \qd ->
            c_render1 ctx pc algorithm 15                       -- the `15` of src/Scene/Trace.hs:200
                      (fromIntegral screenWidth) (fromIntegral screenHeight) nullPtr nullPtr
                      pr pg pb pa pbb pcc pd qr qg qb qa qbb qcc qd
    check ctx rc
    fromPlanes <$> ((,,,,,,) <$> V.unsafeFreeze r' <*> V.unsafeFreeze g' <*> V.unsafeFreeze b'
                              <*> V.unsafeFreeze sa' <*> V.unsafeFreeze sb' <*> V.unsafeFreeze sc' <*> V.unsafeFreeze sd')

download :: Ptr PtmiCtx -> IO RenderResult
download ctx = do
  [r, g, b]        <- mapM (const $ VM.new nPixels) [1 .. 3 :: Int]
  [sa, sb, sc, sd] <- mapM (const $ VM.new nPixels) [1 .. 4 :: Int]
  rc <- VM.unsafeWith r $ \qr -> VM.unsafeWith g $ \qg -> VM.unsafeWith b $ \qb ->
        VM.unsafeWith sa $ \qa -> VM.unsafeWith sb $ \qbb -> VM.unsafeWith sc $ \qcc -> VM.unsafeWith sd $ \qd ->
          c_download ctx qr qg qb qa qbb qcc qd
  check ctx rc
  fromPlanes <$> ((,,,,,,) <$> V.unsafeFreeze r <*> V.unsafeFreeze g <*> V.unsafeFreeze b
                            <*> V.unsafeFreeze sa <*> V.unsafeFreeze sb <*> V.unsafeFreeze sc <*> V.unsafeFreeze sd)

-- | @run <$> initialOutput@ (src/Util.hs:204-205): zero colour + fresh RNG states, generated on the device.
-- The reference seeds from OS entropy (src/Util.hs:122-127); pass any 64-bit seed (e.g. drawn from
-- System.Random.MWC.createSystemRandom) -- equal seeds give equal images.
initialOutput :: Handle -> Word64 -> IO RenderResult
initialOutput (Handle fp) seed0 = withForeignPtr fp $ \ctx -> c_init_output ctx seed0 >>= check ctx >> download ctx

-- | @run <$> reseed acc@ (src/Util.hs:134-135): keep the colour, replace every RNG state.
reseed :: Handle -> Word64 -> RenderResult -> IO RenderResult
reseed (Handle fp) seed0 acc = withForeignPtr fp $ \ctx -> do
  let (r, g, b, _, _, _, _) = planesOf acc
  rc <- V.unsafeWith r $ \pr -> V.unsafeWith g $ \pg -> V.unsafeWith b $ \pb ->
        c_upload ctx pr pg pb nullPtr nullPtr nullPtr nullPtr
  check ctx rc
  c_reseed ctx seed0 >>= check ctx
  download ctx

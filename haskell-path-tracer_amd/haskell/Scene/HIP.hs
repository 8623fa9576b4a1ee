{-# LANGUAGE ForeignFunctionInterface #-}
{-# LANGUAGE BangPatterns #-}
{-# LANGUAGE RecordWildCards #-}

-- | Replacement for the Accelerate-compiled render function of haskell-path-tracer on libptmi (include/ptmi.h).
--
-- STATUS: SOURCE ONLY, NEVER COMPILED.  The build container and the GPU boxes have no GHC, so no line of this module
-- has been type-checked; the C++ mirror of the same flows (hostcxx/scene.hpp) is what the tests run.  It uses nothing
-- beyond @base@, @vector@ and the packages the application already depends on (accelerate, accelerate-io-vector, linear).
--
-- Two wirings are offered:
--
-- 1. RESIDENT (recommended).  The accumulator and the RNG planes stay on the device; one call renders a whole batch
--    of samples; only the three colour planes graphicsLoop shows ever cross PCIe.  In app/Main.hs:
--
--    > hip <- HIP.initialise 0 screenWidth screenHeight           -- once in main, before :154 (sizes are run-time values here)
--    > HIP.resetOutput hip seed0                                   -- replaces `run <$> initialOutput`            (:155, :306)
--    > -- computationLoop (:199-215): `doTimes batchSize (compute camera) value` becomes ONE device call
--    > iterations' <- HIP.renderResident hip config camera batchSize iterations    -- = iterations + batchSize
--    > HIP.reseedResident hip seed0'                               -- replaces `run <$> (reseed . A.use $ acc)`  (:231)
--    > -- graphicsLoop (:346-351): `A.toVectors texture` + `V.zipWith3 V3` become
--    > (r, g, b) <- HIP.downloadColor hip          -- the three planes, or
--    > rgb       <- HIP.presentRGB32F hip iterations   -- interleaved and already divided by the iteration count (fs.glsl:12)
--
--    The 'Result' record then carries the iteration count only; 'Handle' is the accumulator.  patches/Main.resident.diff is this wiring
--    as a patch against app/Main.hs.
--
-- 2. COMPATIBLE.  'compileFor' builds the very closure of app/Main.hs:188-191,
--    @Camera -> (Int, RenderResult) -> (Int, RenderResult)@, one sample per call, pure -- and, since libptmi 0.6, with the
--    device residency runN has on the reference's own GPU backend: the result of a call STAYS ON THE DEVICE (its seven
--    planes are thunks that download on first use: graphicsLoop forces r, g and b, nobody forces the RNG planes), and a call
--    whose argument is such a result finds it there (ptmi_render1_chained: no upload).  A RenderResult that came from
--    anywhere else is uploaded as before.  800x600: ~30 us per call instead of ~1.1 ms (profiles/r06_c0_calls.json).
--
--    > let compute' = HIP.compileFor hip arguments                 -- replaces `compileFor arguments`            (:154)
--    > seeds <- HIP.initialOutput hip seed0                        -- replaces `run <$> initialOutput`           (:155, :306)
--    > reseeded <- HIP.reseed hip seed0' acc                       -- replaces `run <$> (reseed . A.use $ acc)`  (:231)
--
--    See note [values on the device] for what this rests on.  'compileForCopying' is the closure on ptmi_render1 (seven host
--    planes in, seven out, every call), kept for comparisons with the Accelerate run.
module Scene.HIP
  ( Handle
  , initialise
  , handleSize
  , shutdown
  , libraryBuildId
  -- * resident wiring
  , resetOutput
  , renderResident
  , reseedResident
  , downloadColor
  , presentRGB32F
  , synchronize
  -- * compatible wiring
  , compileFor
  , compileForCopying
  , initialOutput
  , reseed
  , chainStats
  , PtmiError(..)
  ) where

import           Control.Concurrent.MVar
import           Control.Exception
import           Control.Monad                  ( when )
import qualified Data.Array.Accelerate         as A
import           Data.Array.Accelerate.IO.Data.Vector.Storable
                                                ( fromVectors, toVectors )
import qualified Data.Vector.Storable          as V
import qualified Data.Vector.Storable.Mutable  as VM
import           Data.IORef
import           Data.Int
import           Data.Word
import           Foreign
import           Foreign.C.String
import           Foreign.C.Types
import           GHC.Float                      ( castWord32ToFloat )
import           Linear                         ( V3(..) )
import           System.IO.Unsafe               ( unsafeInterleaveIO
                                                , unsafePerformIO
                                                )
import           System.Mem                     ( performMinorGC )
import           System.Mem.StableName

import           Scene.Objects
import           Scene.Trace                    ( Algorithm(..) )
import           Scene.World                    ( mainScene' )   -- see note [scene as data]

-- Note [scene as data]
-- Scene.World.mainScene is a list of Exp constants baked into the Accelerate kernel (src/Scene/World.hs:15-77).
-- libptmi takes the scene at run time, so World.hs additionally exports the plain Haskell values it already contains:
--
-- > mainScene' :: ([Sphere], [Plane])
-- > mainScene' = (spheres', planes')   -- the two where-bound lists, lifted to top level
--
-- Note [image size at run time]
-- The reference fixes the image size at compile time (src/Util.hs:185-188, the author's own TODO).  This module takes
-- width and height as arguments of 'initialise' and carries them in the 'Handle'; nothing here imports
-- Util.screenWidth / screenHeight.  The application may keep passing its constants, or a command-line value.

data PtmiCtx
-- | The device context plus the image size it was created for, and -- for the compatible wiring -- which RenderResults are
-- states of the context (note [values on the device]).  The context lives as long as the process, like the reference's PTX
-- context: it has no finalizer (one could run before the last result's at exit); 'shutdown' destroys it explicitly.
data Handle = Handle
  { handleCtx    :: !(ForeignPtr PtmiCtx)
  , handleWidth  :: !Int
  , handleHeight :: !Int
  , handleHeld   :: !(MVar [(StableName RenderResult, Word64)])   -- results on the device: the array's identity -> its token
  , handleCalls  :: !(IORef Int)                                    -- closure calls since the last minor collection
  }

handleSize :: Handle -> (Int, Int)
handleSize h = (handleWidth h, handleHeight h)

-- | Which code the loaded libptmi holds (ptmi_build_id: a hash over its compiled host and gfx950 code): print it next to any timing,
-- so that a number can be tied to a binary.
libraryBuildId :: IO String
libraryBuildId = c_build_id >>= peekCString

data PtmiError = PtmiError Int String deriving Show
instance Exception PtmiError

-- Calls that run for milliseconds to seconds are `safe`, so that the graphics and input threads keep running
-- (app/Main.hs:178-180).  An address import takes no safety annotation.
foreign import ccall unsafe "ptmi.h ptmi_build_id" c_build_id    :: IO CString
foreign import ccall safe "ptmi.h ptmi_create"      c_create      :: Ptr (Ptr PtmiCtx) -> CInt -> IO CInt
foreign import ccall safe "ptmi.h ptmi_destroy"     c_destroy     :: Ptr PtmiCtx -> IO ()
foreign import ccall safe "ptmi.h ptmi_last_error"  c_last_error  :: Ptr PtmiCtx -> IO CString
foreign import ccall safe "ptmi.h ptmi_set_scene"   c_set_scene   :: Ptr PtmiCtx -> Ptr Float -> CInt -> Ptr Float -> CInt -> IO CInt
foreign import ccall safe "ptmi.h ptmi_resize"      c_resize      :: Ptr PtmiCtx -> CInt -> CInt -> IO CInt
foreign import ccall safe "ptmi.h ptmi_init_output" c_init_output :: Ptr PtmiCtx -> Word64 -> IO CInt
foreign import ccall safe "ptmi.h ptmi_reseed"      c_reseed      :: Ptr PtmiCtx -> Word64 -> IO CInt
foreign import ccall safe "ptmi.h ptmi_render"      c_render      :: Ptr PtmiCtx -> Ptr CamRec -> CInt -> CInt -> CInt -> IO CInt
foreign import ccall safe "ptmi.h ptmi_synchronize" c_synchronize :: Ptr PtmiCtx -> IO CInt
foreign import ccall safe "ptmi.h ptmi_download_color" c_download_color :: Ptr PtmiCtx -> Ptr Float -> Ptr Float -> Ptr Float -> IO CInt
foreign import ccall safe "ptmi.h ptmi_present"     c_present     :: Ptr PtmiCtx -> CInt -> Ptr Float -> Ptr Word8 -> IO CInt
foreign import ccall safe "ptmi.h ptmi_render1"     c_render1
  :: Ptr PtmiCtx -> Ptr CamRec -> CInt -> CInt -> CInt -> CInt -> Ptr Int64 -> Ptr Int64
  -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> Ptr Word32
  -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> Ptr Word32
  -> IO CInt
-- the closure, chained (include/ptmi.h): states under tokens
foreign import ccall safe "ptmi.h ptmi_render1_chained" c_render1_chained
  :: Ptr PtmiCtx -> Ptr CamRec -> CInt -> CInt -> CInt -> CInt -> Word64 -> CInt
  -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> Ptr Word32
  -> Ptr Word64
  -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> Ptr Word32
  -> IO CInt
foreign import ccall safe "ptmi.h ptmi_chain_init_output" c_chain_init_output :: Ptr PtmiCtx -> CInt -> CInt -> Word64 -> Ptr Word64 -> IO CInt
foreign import ccall safe "ptmi.h ptmi_chain_reseed" c_chain_reseed
  :: Ptr PtmiCtx -> Word64 -> CInt -> CInt -> Word64 -> CInt -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Word64 -> IO CInt
foreign import ccall safe "ptmi.h ptmi_chain_fetch" c_chain_fetch
  :: Ptr PtmiCtx -> Word64 -> Ptr Float -> Ptr Float -> Ptr Float -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> Ptr Word32 -> IO CInt
foreign import ccall safe "ptmi.h ptmi_chain_release" c_chain_release :: Ptr PtmiCtx -> Word64 -> IO CInt
foreign import ccall safe "ptmi.h ptmi_chain_info" c_chain_info :: Ptr PtmiCtx -> Ptr Word64 -> IO CInt

-- | PTMI_ESTALE: a token names no state the context holds.
eStale :: CInt
eStale = -7

-- | struct ptmi_camera { float position[3]; float rotation[3]; int64_t fov; }  (32 bytes)
data CamRec
pokeCamera :: Ptr CamRec -> Camera -> IO ()
pokeCamera p Camera{..} = do
  let V3 px py pz = _cameraPosition
      V3 rx ry rz = _cameraRotation
  pokeArray (castPtr p) [px, py, pz, rx, ry, rz :: Float]
  pokeByteOff p 24 (fromIntegral _cameraFov :: Int64)

-- (ptmi_last_error hands out the calling OS thread's own message, or -- an unbound Haskell thread may have moved between the two calls -- a
-- copy of the context's latest one: never a pointer another thread's failure can free.)
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

algorithmTag :: Algorithm -> CInt
algorithmTag Streams = 0                        -- PTMI_STREAMS
algorithmTag Inline  = 1                        -- PTMI_INLINE

-- | The `15` of src/Scene/Trace.hs:200 / maxIterations (:80-81).
bounceLimit :: CInt
bounceLimit = 15

-- | Create the context on a device, upload mainScene and size it to @width x height@ (run-time values; the
-- reference's constants are src/Util.hs:186-188).
initialise :: Int -> Int -> Int -> IO Handle
initialise device width height = alloca $ \pp -> do
  rc <- c_create pp (fromIntegral device)
  when (rc /= 0) $ c_last_error nullPtr >>= peekCString >>= throwIO . PtmiError (fromIntegral rc)
  ctx <- peek pp
  fp  <- newForeignPtr_ ctx                       -- no finalizer: see 'Handle'
  let (spheres, planes) = mainScene'
  -- the records travel as raw 32-bit words (the BRDF tag is an int32 bit pattern in a float slot): no conversion
  withArray (concatMap sphereWords spheres) $ \ps ->
    withArray (concatMap planeWords planes) $ \pp' ->
      c_set_scene ctx ps (fromIntegral $ length spheres) pp' (fromIntegral $ length planes) >>= check ctx
  c_resize ctx (fromIntegral width) (fromIntegral height) >>= check ctx
  held  <- newMVar []
  calls <- newIORef 0
  return (Handle fp width height held calls)

-- | Destroy the context.  No other function of this module may be used on the handle afterwards, and no RenderResult made by
-- 'compileFor' / 'initialOutput' / 'reseed' whose planes have not been read yet may be read.
shutdown :: Handle -> IO ()
shutdown h = withForeignPtr (handleCtx h) c_destroy

nPixels :: Handle -> Int
nPixels h = handleWidth h * handleHeight h

-- ---------------------------------------------------------------------------------------------------------------------
-- Resident wiring
-- ---------------------------------------------------------------------------------------------------------------------

-- | @run <$> initialOutput@ (src/Util.hs:204-205; app/Main.hs:155, :306) without the read-back: colour := 0, RNG :=
-- genSeeds, on the device.  The reference seeds from OS entropy (src/Util.hs:122-127); pass any 64-bit seed (e.g. drawn
-- from System.Random.MWC.createSystemRandom) -- equal seeds give equal images.
resetOutput :: Handle -> Word64 -> IO ()
resetOutput h seed0 = withForeignPtr (handleCtx h) $ \ctx -> c_init_output ctx seed0 >>= check ctx

-- | @doTimes n (compute camera)@ (app/Main.hs:208-211, :241-242) as ONE device call: n successive applications of
-- @render algorithm screenPixels camera@ to the resident accumulator.  Returns the new iteration count.  The launch is
-- asynchronous; the next 'downloadColor' / 'presentRGB32F' / 'synchronize' waits for it.
renderResident :: Handle -> Algorithm -> Camera -> Int -> Int -> IO Int
renderResident h config cam n iterations = withForeignPtr (handleCtx h) $ \ctx -> allocaBytes 32 $ \pc -> do
  pokeCamera pc cam
  c_render ctx pc (algorithmTag config) bounceLimit (fromIntegral n) >>= check ctx
  return (iterations + n)

-- | @run <$> reseed acc@ (src/Util.hs:134-135; app/Main.hs:231) without moving the colour: every RNG state replaced.
reseedResident :: Handle -> Word64 -> IO ()
reseedResident h seed0 = withForeignPtr (handleCtx h) $ \ctx -> c_reseed ctx seed0 >>= check ctx

synchronize :: Handle -> IO ()
synchronize h = withForeignPtr (handleCtx h) $ \ctx -> c_synchronize ctx >>= check ctx

-- | What graphicsLoop takes out of the result (app/Main.hs:346-350): the r, g and b planes, a SUM over the samples.
downloadColor :: Handle -> IO (V.Vector Float, V.Vector Float, V.Vector Float)
downloadColor h = withForeignPtr (handleCtx h) $ \ctx -> do
  [r, g, b] <- mapM (const $ VM.new (nPixels h)) [1 .. 3 :: Int]
  rc <- VM.unsafeWith r $ \qr -> VM.unsafeWith g $ \qg -> VM.unsafeWith b $ \qb -> c_download_color ctx qr qg qb
  check ctx rc
  (,,) <$> V.unsafeFreeze r <*> V.unsafeFreeze g <*> V.unsafeFreeze b

-- | The texture graphicsLoop uploads (app/Main.hs:351, :383-393) with the shader's division already applied
-- (app/assets/fs.glsl:12): interleaved RGB32F, colour / iterations, produced on the device.
presentRGB32F :: Handle -> Int -> IO (V.Vector Float)
presentRGB32F h iterations = withForeignPtr (handleCtx h) $ \ctx -> do
  rgb <- VM.new (3 * nPixels h)
  rc  <- VM.unsafeWith rgb $ \q -> c_present ctx (fromIntegral iterations) q nullPtr
  check ctx rc
  V.unsafeFreeze rgb

-- ---------------------------------------------------------------------------------------------------------------------
-- Compatible wiring
-- ---------------------------------------------------------------------------------------------------------------------

type CompiledFunction = Camera -> (Int, RenderResult) -> (Int, RenderResult)

-- | The seven planes of a RenderResult, in libptmi's order (r, g, b, a, b, c, counter).
--
-- ASSUMPTION (unverified, see DESIGN.md section 2, A3): Accelerate represents @Matrix (V3 Float, SFC32)@ as nested pairs
-- of vectors -- the colour part is visible at app/Main.hs:350, @(((), ((((), r), g), b)), seedPlanes)@ -- and SFC32 as
-- four Word32 planes nested the same way, in the order (a, b, c, counter) of PractRand's sfc32 state.  The SFC32 type
-- lives in sfc-random-accelerate, which is not in the reference tree; if its Elt representation orders or nests the
-- words differently, only the pattern below changes.
planesOf :: RenderResult -> (V.Vector Float, V.Vector Float, V.Vector Float, V.Vector Word32, V.Vector Word32, V.Vector Word32, V.Vector Word32)
planesOf acc = let (((), ((((), r), g), b)), (((((), sa), sb), sc), sd)) = toVectors acc in (r, g, b, sa, sb, sc, sd)

fromPlanes :: Handle -> (V.Vector Float, V.Vector Float, V.Vector Float, V.Vector Word32, V.Vector Word32, V.Vector Word32, V.Vector Word32) -> RenderResult
fromPlanes h (r, g, b, sa, sb, sc, sd) =
  fromVectors (A.Z A.:. handleHeight h A.:. handleWidth h)         -- screenShape, src/Util.hs:213-214: Z :. height :. width
              (((), ((((), r), g), b)), (((((), sa), sb), sc), sd))

-- Note [values on the device]
-- A result of 'compileFor' (and of 'initialOutput' / 'reseed') is an ordinary @RenderResult@ -- @fromVectors@ of seven storable
-- vectors -- whose vectors are THUNKS: the colour thunk downloads r, g and b together (ptmi_chain_fetch) when one of them is
-- first demanded, the seed thunk the four RNG planes; until then the planes exist on the device only, as a STATE of the context
-- under a token.  To libptmi a state is an immutable value (a later call renders into a copy on the device; a state that has to
-- leave the device is kept in host memory by the library), so the thunks are pure in the sense `unsafePerformIO` needs: whenever
-- they run they give the planes of exactly this result.
--
-- Which arrays are such values is recorded by IDENTITY: 'handleHeld' maps the StableName of the (evaluated) array to its token.
-- The closure looks its argument up there; a hit passes the token and no planes, a miss -- any RenderResult made elsewhere --
-- passes the seven host planes (ptmi_render1's copy path).  Either way the result is the same function of the argument's planes.
--
-- A token is released by a finalizer on an IORef that only the two thunks reference: once the array is garbage (or every plane has
-- been read, after which the host copy serves) libptmi may reuse the state's memory.  Finalizers run after a collection, and this
-- loop allocates little, so every 32 calls the closure runs a minor collection (~0.1 ms); libptmi keeps up to 64 states on the
-- device before the oldest moves to the host, which costs time (one download) but never a result.
--
-- ASSUMPTIONS (unverified -- no GHC here; DESIGN.md section 2, A8):
--   * @fromVectors@ and the @Array@ constructor are lazy in the vectors, as accelerate-io-vector 0.1 / accelerate 1.3 are believed
--     to be (payload = lazy nested pairs).  If they force them, every call downloads its result (no upload still): correct, slower.
--   * the array object handed back to the closure is the one it returned (GHC does not re-box it on the way through
--     @Result@ / @iterate@); a re-boxed array is a registry miss: correct, one download + upload for that call.
-- The RESIDENT wiring rests on neither.

-- | The seven planes of a state as two lazily downloaded groups, the token's finalizer attached.
heldPlanes :: Handle -> Word64 -> IO (V.Vector Float, V.Vector Float, V.Vector Float, V.Vector Word32, V.Vector Word32, V.Vector Word32, V.Vector Word32)
heldPlanes h token = do
  alive <- newIORef token
  _ <- mkWeakIORef alive $ withForeignPtr (handleCtx h) $ \ctx -> do
         modifyMVar_ (handleHeld h) (return . filter ((/= token) . snd))
         () <$ c_chain_release ctx token
  let n = nPixels h
  colour <- unsafeInterleaveIO $ withForeignPtr (handleCtx h) $ \ctx -> do
    [r, g, b] <- mapM (const $ VM.new n) [1 .. 3 :: Int]
    t  <- readIORef alive                                       -- (keeps `alive` reachable from this thunk)
    rc <- VM.unsafeWith r $ \qr -> VM.unsafeWith g $ \qg -> VM.unsafeWith b $ \qb ->
            c_chain_fetch ctx t qr qg qb nullPtr nullPtr nullPtr nullPtr
    check ctx rc
    (,,) <$> V.unsafeFreeze r <*> V.unsafeFreeze g <*> V.unsafeFreeze b
  seeds <- unsafeInterleaveIO $ withForeignPtr (handleCtx h) $ \ctx -> do
    [sa, sb, sc, sd] <- mapM (const $ VM.new n) [1 .. 4 :: Int]
    t  <- readIORef alive
    rc <- VM.unsafeWith sa $ \qa -> VM.unsafeWith sb $ \qbb -> VM.unsafeWith sc $ \qcc -> VM.unsafeWith sd $ \qd ->
            c_chain_fetch ctx t nullPtr nullPtr nullPtr qa qbb qcc qd
    check ctx rc
    (,,,) <$> V.unsafeFreeze sa <*> V.unsafeFreeze sb <*> V.unsafeFreeze sc <*> V.unsafeFreeze sd
  let (r, g, b)        = colour                                 -- lazy patterns: nothing is demanded here
      (sa, sb, sc, sd) = seeds
  return (r, g, b, sa, sb, sc, sd)

-- | A state of the context as a RenderResult, registered under its identity.
heldResult :: Handle -> Word64 -> IO RenderResult
heldResult h token = do
  arr  <- heldPlanes h token >>= evaluate . fromPlanes h       -- WHNF: the Array constructor, not its vectors
  name <- makeStableName arr
  modifyMVar_ (handleHeld h) (return . ((name, token) :))
  return arr

-- | The token an (evaluated) array is held under, if it is a result of this handle.
tokenOf :: Handle -> RenderResult -> IO (Maybe Word64)
tokenOf h acc = do
  name <- makeStableName acc
  lookup name <$> readMVar (handleHeld h)

-- | Run @withToken token@; if libptmi no longer holds the token (PTMI_ESTALE) or the array is not a held one, run
-- @withPlanes@ on the array's host planes (forcing them: a download if they are still thunks of ANOTHER live token, none if they
-- were read before).
chained :: Handle -> Ptr PtmiCtx -> RenderResult -> (Word64 -> IO CInt) -> IO CInt -> IO CInt
chained h _ctx acc withToken withPlanes = do
  held <- tokenOf h acc
  rc   <- maybe (return eStale) withToken held
  if rc == eStale then withPlanes else return rc

-- | @compileFor@ (app/Main.hs:188-191) on libptmi: one call = one sample, exactly `runN (render config) screenPixels`, the
-- result left on the device (note [values on the device]).  The closure is pure from the caller's point of view, like `dewit`.
compileFor :: Handle -> Algorithm -> CompiledFunction
compileFor h !config = \(!c) (!iterations, !acc) ->
  (iterations + 1, unsafePerformIO (render1 c acc))
 where
  w = fromIntegral (handleWidth h)
  hgt = fromIntegral (handleHeight h)
  render1 cam acc = withForeignPtr (handleCtx h) $ \ctx -> allocaBytes 32 $ \pc -> alloca $ \ptok -> do
    pokeCamera pc cam
    let call token pr pg pb pa pbb pcc pd =
          c_render1_chained ctx pc (algorithmTag config) bounceLimit w hgt token 0
                            pr pg pb pa pbb pcc pd ptok
                            nullPtr nullPtr nullPtr nullPtr nullPtr nullPtr nullPtr
    rc <- chained h ctx acc
            (\token -> call token nullPtr nullPtr nullPtr nullPtr nullPtr nullPtr nullPtr)
            (let (r, g, b, sa, sb, sc, sd) = planesOf acc
             in  V.unsafeWith r $ \pr -> V.unsafeWith g $ \pg -> V.unsafeWith b $ \pb ->
                 V.unsafeWith sa $ \pa -> V.unsafeWith sb $ \pbb -> V.unsafeWith sc $ \pcc -> V.unsafeWith sd $ \pd ->
                   call 0 pr pg pb pa pbb pcc pd)
    check ctx rc
    calls <- atomicModifyIORef' (handleCalls h) (\k -> let k' = if k >= 31 then 0 else k + 1 in (k', k))
    when (calls >= 31) performMinorGC                        -- let the finalizers of dead intermediate results release their states
    peek ptok >>= heldResult h

-- | The same closure on ptmi_render1: seven host planes in, seven fresh ones out, every call (56 bytes per pixel over PCIe each
-- way).  What 'compileFor' was until libptmi 0.5; for comparisons.
compileForCopying :: Handle -> Algorithm -> CompiledFunction
compileForCopying h !config = \(!c) (!iterations, !acc) ->
  (iterations + 1, unsafePerformIO (render1 c acc))
 where
  n = nPixels h
  render1 cam acc = withForeignPtr (handleCtx h) $ \ctx -> allocaBytes 32 $ \pc -> do
    pokeCamera pc cam
    let (r, g, b, sa, sb, sc, sd) = planesOf acc
    [r', g', b']        <- mapM (const $ VM.new n) [1 .. 3 :: Int]
    [sa', sb', sc', sd'] <- mapM (const $ VM.new n) [1 .. 4 :: Int]
    rc <- V.unsafeWith r $ \pr -> V.unsafeWith g $ \pg -> V.unsafeWith b $ \pb ->
          V.unsafeWith sa $ \pa -> V.unsafeWith sb $ \pbb -> V.unsafeWith sc $ \pcc -> V.unsafeWith sd $ \pd ->
          VM.unsafeWith r' $ \qr -> VM.unsafeWith g' $ \qg -> VM.unsafeWith b' $ \qb ->
          VM.unsafeWith sa' $ \qa -> VM.unsafeWith sb' $ \qbb -> VM.unsafeWith sc' $ \qcc -> VM.unsafeWith sd' $ \qd ->
            c_render1 ctx pc (algorithmTag config) bounceLimit
                      (fromIntegral $ handleWidth h) (fromIntegral $ handleHeight h) nullPtr nullPtr
                      pr pg pb pa pbb pcc pd qr qg qb qa qbb qcc qd
    check ctx rc
    fromPlanes h <$> ((,,,,,,) <$> V.unsafeFreeze r' <*> V.unsafeFreeze g' <*> V.unsafeFreeze b'
                                <*> V.unsafeFreeze sa' <*> V.unsafeFreeze sb' <*> V.unsafeFreeze sc' <*> V.unsafeFreeze sd')

-- | @run <$> initialOutput@ (src/Util.hs:204-205): colour 0 and fresh RNG states, made on the device and left there.
initialOutput :: Handle -> Word64 -> IO RenderResult
initialOutput h seed0 = withForeignPtr (handleCtx h) $ \ctx -> alloca $ \ptok -> do
  c_chain_init_output ctx (fromIntegral $ handleWidth h) (fromIntegral $ handleHeight h) seed0 ptok >>= check ctx
  peek ptok >>= heldResult h

-- | @run <$> reseed acc@ (src/Util.hs:134-135): keep the colour, replace every RNG state -- on the device when @acc@ is there,
-- else from its colour planes.
reseed :: Handle -> Word64 -> RenderResult -> IO RenderResult
reseed h seed0 acc0 = withForeignPtr (handleCtx h) $ \ctx -> alloca $ \ptok -> do
  acc <- evaluate acc0
  let w = fromIntegral (handleWidth h)
      hgt = fromIntegral (handleHeight h)
  rc <- chained h ctx acc
          (\token -> c_chain_reseed ctx seed0 w hgt token 0 nullPtr nullPtr nullPtr ptok)
          (let (r, g, b, _, _, _, _) = planesOf acc
           in  V.unsafeWith r $ \pr -> V.unsafeWith g $ \pg -> V.unsafeWith b $ \pb ->
                 c_chain_reseed ctx seed0 w hgt 0 0 pr pg pb ptok)
  check ctx rc
  peek ptok >>= heldResult h

-- | (states on the device, states moved to the host, calls that found their input on the device, calls that uploaded it,
-- evictions) -- struct ptmi_chain_stats; print it to see whether the closure runs as intended.
chainStats :: Handle -> IO (Int, Int, Int, Int, Int)
chainStats h = withForeignPtr (handleCtx h) $ \ctx -> allocaBytes 64 $ \p -> do
  c_chain_info ctx (castPtr p) >>= check ctx
  onDevice <- peekByteOff p 0 :: IO Word32
  onHost   <- peekByteOff p 4 :: IO Word32
  found    <- peekByteOff p 24 :: IO Word64
  uploaded <- peekByteOff p 40 :: IO Word64
  evicted  <- peekByteOff p 48 :: IO Word64
  return (fromIntegral onDevice, fromIntegral onHost, fromIntegral found, fromIntegral uploaded, fromIntegral evicted)
